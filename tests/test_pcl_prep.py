"""Row f3: point-cloud preparation.  The golden holds what the REFERENCE functions (crop_ball_from_depth_image,
crop_mask_depth_image on backproject_th's map) return per instance of a synthetic depth frame, plus the torch.randperm
draws they consumed.  Index / selection work is compared exactly; the 3-D points to 1e-6 abs (the kernel multiplies
and divides in the reference's order, differences are fp32 rounding of the same expression)."""
import os

import numpy as np
import pytest
import torch

from oracle import pcl_oracle as PO
from tests.util import GOLDEN_DIR

DEV = "cuda:0"


def _g():
    z = np.load(os.path.join(GOLDEN_DIR, "pcl_prep.npz"))
    return {k: z[k] for k in z.files}


def _scene(g, dev="cpu"):
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    return t("in_depth"), t("in_K"), t("in_masks"), t("in_poses"), t("in_scales")


def test_pcl_oracle_matches_reference_functions():
    g = _g()
    depth, K, masks, poses, scales = _scene(g)
    N = int(g["meta"][0])
    for mode, ball in (("ball", True), ("mask", False)):
        for i in range(len(masks)):
            pix, bp = PO.candidates(depth, K, masks[i], poses[i], scales[i], 0.5, use_ball=ball)
            assert len(pix) == g[f"{mode}_counts"][i]
            got, _ = PO.sample(pix, bp, torch.from_numpy(g[f"{mode}_sample_idx"][i]))
            assert np.abs(got.numpy() - g[f"{mode}_pcl"][i]).max() < 1e-7
    # the scene exercises every branch of the radius search
    c = g["ball_counts"]
    assert c[1] < c[0] and c[3] == g["mask_counts"][3] and c[4] < N


def test_fps_oracle_matches_reference_farthest_point_sampling():
    """INPUT.FPS_SAMPLE: the golden holds crop_ball_from_depth_image(..., device="cpu", fps_sample=True) of the
    reference; the scene covers plain lists, lists tiled x2 (61 -> 122) and x16 (11 -> 176: duplicated points, i.e.
    exact distance ties that only "first maximum wins" resolves like torch.argmax)."""
    g = _g()
    depth, K, masks, poses, scales = _scene(g)
    N = int(g["meta"][0])
    for i in range(len(masks)):
        pix, bp = PO.candidates(depth, K, masks[i], poses[i], scales[i], 0.5, use_ball=True)
        s = PO.fps_sample_idx(pix, bp, N)
        assert s.tolist() == g["fps_sample_idx"][i].tolist()
        got, _ = PO.sample(pix, bp, s)
        assert np.abs(got.numpy() - g["fps_pcl"][i]).max() < 1e-7


@pytest.mark.gpu
def test_hip_fps_sampling_matches_reference():
    """Device farthest point sampling (k_pcl_fps) picks the reference's points in the reference's order."""
    from catre_amd import pcl_prep

    g = _g()
    depth, K, masks, poses, scales = _scene(g, DEV)
    N = int(g["meta"][0])
    pcl, pix, counts = pcl_prep.sample_instances(depth, K, masks, poses, scales, ratio=0.5, num_points=N, use_ball=True,
                                                 fps_sample=True, return_pixels=True)
    assert counts.cpu().tolist() == g["ball_counts"].tolist()
    assert np.abs(pcl.cpu().numpy() - g["fps_pcl"]).max() < 1e-6
    # the single-instance signature of the reference
    rgb, pts, nocs = pcl_prep.crop_ball_from_depth_image(None, depth, masks[0], poses[0], scales[0], 0.5, K, num_points=N,
                                                         fps_sample=True)
    assert np.abs(pts.cpu().numpy() - g["fps_pcl"][0]).max() < 1e-6
    # a full-size frame: distinct pixels while the list is long enough, deterministic, and greedy-farthest
    from catre_amd import synth

    sc = synth.make_depth_scene(seed=5, H=480, W=640, n_inst=3)
    d, Kb = sc["depth"].to(DEV), sc["K"]
    m, p, s = sc["masks"].to(DEV), sc["poses"].to(DEV), sc["scales"].to(DEV)
    a, pa, ca = pcl_prep.sample_instances(d, Kb, m, p, s, num_points=1024, fps_sample=True, return_pixels=True)
    b, _, _ = pcl_prep.sample_instances(d, Kb, m, p, s, num_points=1024, fps_sample=True, return_pixels=True)
    assert torch.equal(a, b)
    for i, c in enumerate(ca.cpu().tolist()):
        if c >= 1024:
            assert len(set(pa[i].cpu().tolist())) == 1024
        pts_i = a[i].cpu().double()
        # pick k+1 is the farthest remaining point from picks 0..k: its distance to them is >= every later pick's
        dmin = torch.cdist(pts_i[1:9], pts_i[:1]).min(1)[0]
        assert dmin[0] >= dmin[1:].max() - 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ball", "mask"])
def test_hip_pcl_prep_matches_reference_with_its_random_draws(mode):
    from catre_amd import hip, pcl_prep

    g = _g()
    depth, K, masks, poses, scales = _scene(g, DEV)
    N, seed = int(g["meta"][0]), int(g["meta"][1])
    ball = mode == "ball"
    torch.manual_seed(seed)  # the golden's draws: one permutation per instance (mask crop: more for a short list)
    pcl, pix, counts = pcl_prep.sample_instances(depth, K, masks, poses, scales, ratio=0.5, num_points=N,
                                                 use_ball=ball, sample="host", return_pixels=True)
    assert counts.cpu().tolist() == g[f"{mode}_counts"].tolist()
    want = g[f"{mode}_pcl"]
    assert np.abs(pcl.cpu().numpy() - want).max() < 1e-6
    # the pixel indices point at the returned points
    bp = PO.backproject(depth.cpu(), K.cpu()).reshape(-1, 3)
    assert np.abs(bp[pix.cpu().long()].numpy() - pcl.cpu().numpy()).max() < 1e-6


@pytest.mark.gpu
def test_hip_candidate_lists_are_the_reference_sets_in_nonzero_order():
    from catre_amd import hip

    g = _g()
    depth, K, masks, poses, scales = _scene(g, DEV)
    I, (H, W) = len(masks), depth.shape
    lib = hip.load()
    nbytes = lib.catre_pcl_workspace_bytes(I, H, W)
    ws = torch.zeros(nbytes // 4, dtype=torch.int32, device=DEV)
    counts = torch.zeros(I, dtype=torch.int32, device=DEV)
    import ctypes
    k9 = (ctypes.c_float * 9)(*[float(v) for v in K.cpu().reshape(-1)])
    m8 = masks.to(torch.uint8).contiguous()
    hip.check(lib.catre_pcl_candidates(hip.ptr(depth), k9, hip.ptr(m8), hip.ptr(poses.contiguous()),
                                       hip.ptr(scales.contiguous()), 0.5, 1, I, H, W, hip.ptr(ws), nbytes, hip.ptr(counts),
                                       hip.stream_ptr(depth.device)), "catre_pcl_candidates")
    cand = ws[-(((I * H * W + 63) // 64) * 64):][: I * H * W].reshape(I, H * W).cpu()
    for i in range(I):
        want, _ = PO.candidates(depth.cpu(), K.cpu(), masks[i].cpu(), poses[i].cpu(), scales[i].cpu(), 0.5)
        n = int(counts[i])
        assert n == len(want) and torch.equal(cand[i, :n].long(), want), i


@pytest.mark.gpu
def test_hip_device_sampling_properties_and_full_frame():
    """sample='device': without replacement whenever there are enough candidates, always inside the candidate set,
    deterministic per seed, different across seeds and instances; and a 480x640 frame with 8 instances."""
    from catre_amd import pcl_prep, synth

    g = _g()
    depth, K, masks, poses, scales = _scene(g, DEV)
    N = int(g["meta"][0])
    a, pa, counts = pcl_prep.sample_instances(depth, K, masks, poses, scales, num_points=N, sample="device", seed=5,
                                              return_pixels=True)
    b, pb, _ = pcl_prep.sample_instances(depth, K, masks, poses, scales, num_points=N, sample="device", seed=5,
                                         return_pixels=True)
    c, pc, _ = pcl_prep.sample_instances(depth, K, masks, poses, scales, num_points=N, sample="device", seed=6,
                                         return_pixels=True)
    assert torch.equal(a, b) and torch.equal(pa, pb) and not torch.equal(pa, pc)
    for i in range(len(masks)):
        want, _ = PO.candidates(depth.cpu(), K.cpu(), masks[i].cpu(), poses[i].cpu(), scales[i].cpu(), 0.5)
        got = pa[i].cpu().long()
        assert set(got.tolist()) <= set(want.tolist())
        n = len(want)
        if n >= N:
            assert len(set(got.tolist())) == N, "sampling without replacement"
        else:  # tiled list: every candidate appears, none more than ceil(L/n) times
            L = PO.tiled_length(n, N)
            assert np.bincount(got.numpy(), minlength=1).max() <= L // n
    sc = synth.make_depth_scene(H=480, W=640, n_inst=8, seed=3)
    d, Kf = sc["depth"].to(DEV), sc["K"]
    pcl, pix, cnt = pcl_prep.sample_instances(d, Kf, sc["masks"].to(DEV), sc["poses"].to(DEV), sc["scales"].to(DEV),
                                              num_points=1024, sample="device", seed=1, return_pixels=True)
    assert pcl.shape == (8, 1024, 3) and torch.isfinite(pcl).all() and (pcl[..., 2] > 0).all()
    for i in range(8):
        want, _ = PO.candidates(sc["depth"], Kf, sc["masks"][i], sc["poses"][i], sc["scales"][i], 0.5)
        assert int(cnt[i]) == len(want) and set(pix[i].cpu().tolist()) <= set(want.tolist())
    # reference-style single-instance call
    torch.manual_seed(1)
    _, pts, _ = pcl_prep.crop_ball_from_depth_image(None, d, sc["masks"][0].to(DEV), sc["poses"][0].to(DEV),
                                                    sc["scales"][0].to(DEV), 0.5, Kf, num_points=256)
    torch.manual_seed(1)
    pix0, bp = PO.candidates(sc["depth"], Kf, sc["masks"][0], sc["poses"][0], sc["scales"][0], 0.5)
    ref, _ = PO.sample(pix0, bp, torch.randperm(PO.tiled_length(len(pix0), 256))[:256])
    assert np.abs(pts.cpu().numpy() - ref.numpy()).max() < 1e-6
    empty = torch.zeros(1, 480, 640, dtype=torch.bool, device=DEV)
    with pytest.raises(ValueError):
        pcl_prep.sample_instances(d, Kf, empty, sc["poses"][:1].to(DEV), sc["scales"][:1].to(DEV), sample="host")
    z, pz, cz = pcl_prep.sample_instances(d, Kf, empty, sc["poses"][:1].to(DEV), sc["scales"][:1].to(DEV), num_points=8,
                                          sample="device", return_pixels=True)
    assert int(cz[0]) == 0 and (pz == -1).all() and (z == 0).all()
