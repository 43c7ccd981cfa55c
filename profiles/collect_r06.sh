#!/bin/bash
# Everything the round-6 docs quote, collected on a GPU box from the build of this commit (run from the repo root), in
# parts that each finish within a few minutes and write only small summaries:
#   profiles/collect_r06.sh tests|bench|small|prof|train|launches|ddp|pmc|pmctrain|bf16|sweep      -> gpurun_out/r06_*   (copy the summaries into profiles/)
o=gpurun_out
part=${1:-all}
want() { [ "$part" = all ] || [ "$part" = "$1" ]; }
if want tests; then
  python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^[0-9]+ passed|^[0-9]+ failed| passed| failed|^ERROR|^FAILED" | tail -8 > $o/r06_gputest.log
  cat $o/r06_gputest.log
fi
if want bench; then
  python bench.py > $o/r06_bench_n1.json 2> $o/r06_bench_n1.err
  { python bench.py --dtype split --no-cpu-baseline; python bench.py --dtype bf16 --no-cpu-baseline;
    python bench.py --shape config5 --no-cpu-baseline --steps 5; python bench.py --shape config5 --dtype bf16 --no-cpu-baseline --steps 5;
    python bench.py --shape config5 --dtype split --no-cpu-baseline --steps 5; } 2>/dev/null | grep '^{' > $o/r06_modes_bench.jsonl
  { python bench.py --mode train --steps 4 --warmup 4; python bench.py --mode train --dtype bf16 --steps 4 --warmup 4;
    python bench.py --mode train --dtype split --steps 4 --warmup 4; } 2>/dev/null | grep '^{' > $o/r06_train_bench.jsonl
fi
if want small; then
  python profiles/small_batch.py 2>/dev/null | grep '^{' > $o/r06_other_configs.jsonl
  python profiles/small_sweep.py 1 2 3 4 6 8 16 32 2>/dev/null | grep '^{' > $o/r06_small_sweep.jsonl
  python profiles/eval_loop_probe.py 1 2 4 6 2>/dev/null | grep '^{' > $o/r06_eval_loop.jsonl
  { python profiles/multi_stream_probe.py 1 1 2 4 8; python profiles/multi_stream_probe.py 4 1 2 4; } 2>/dev/null | grep '^{' > $o/r06_multi_stream.jsonl
  profiles/prof.sh $o/r06_b1_kernel_stats.csv python $PWD/profiles/b1_profile.py 1 50
fi
if want prof; then
  profiles/prof.sh $o/r06_kernel_stats.csv python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra
fi
if want train; then
  python profiles/train_step.py 16 64 256 2>/dev/null | grep '^{' > $o/r06_train_step.jsonl
  PROF_TRACE="$PWD/$o/r06_train_trace.csv 1500" profiles/prof.sh $o/r06_train_kernel_stats.csv python $PWD/bench.py --mode train --steps 3 --warmup 1
  python profiles/gemm_probe.py 2>/dev/null | grep '^{' > $o/r06_gemm_probe.jsonl
  profiles/prof.sh $o/r06_train_bf16_kernel_stats.csv python $PWD/bench.py --mode train --dtype bf16 --steps 3 --warmup 1
  profiles/prof.sh $o/r06_train_split_kernel_stats.csv python $PWD/bench.py --mode train --dtype split --steps 3 --warmup 1
fi
if want launches; then   # what an iteration launches and from where, where the GPU idles, how far ahead the host runs
  python profiles/train_launch_census.py 2>/dev/null | grep -v "^\[W\|Warning\|warn" > $o/r06_train_launch_census.txt
  AMP=1 python profiles/train_launch_census.py 2>/dev/null | grep -v "^\[W\|Warning\|warn" > $o/r06_train_launch_census_amp.txt
  PROF_TRACE="$PWD/$o/r06_train_trace.csv 1500" profiles/prof.sh /tmp/r06_tk.csv python $PWD/bench.py --mode train --steps 3 --warmup 1 > /dev/null
  { echo "fp32:"; python profiles/gap_report.py $o/r06_train_trace.csv 8; } > $o/r06_train_gaps.txt
  python profiles/iteration_stats.py $o/r06_train_trace.csv $o/r06_train_iter_kernel_stats.csv > $o/r06_train_iter_counts.txt
  PROF_TRACE="$PWD/$o/r06_train_bf16_trace.csv 1500" profiles/prof.sh /tmp/r06_tk.csv python $PWD/bench.py --mode train --dtype bf16 --steps 3 --warmup 1 > /dev/null
  { echo "autocast (under rocprofv3 the host, not the GPU, paces this mode - r06_train_host_time.jsonl has the unprofiled margins):"; python profiles/gap_report.py $o/r06_train_bf16_trace.csv 8; } >> $o/r06_train_gaps.txt
  python profiles/iteration_stats.py $o/r06_train_bf16_trace.csv $o/r06_train_bf16_iter_kernel_stats.csv >> $o/r06_train_iter_counts.txt
  { python profiles/train_host_time.py 256; AMP=1 python profiles/train_host_time.py 256; python profiles/train_host_time.py 16; AMP=1 python profiles/train_host_time.py 16; } 2>/dev/null | grep '^{' > $o/r06_train_host_time.jsonl
  rm -f $o/r06_train_trace.csv $o/r06_train_bf16_trace.csv /tmp/r06_tk.csv
fi
if want ddp; then   # the reference's DDP wrap on a one-rank RCCL group: the default line's block on a second box, 4 MiB buckets + view, kernels
  python bench.py > $o/r06_bench_ddp.json 2> /dev/null
  python bench.py --mode train --ddp-world1 --bucket-cap-mb 4 --bucket-view --steps 4 --warmup 2 2>/dev/null | grep '^{' > $o/r06_train_ddp4.json
  profiles/prof.sh $o/r06_train_ddp_kernel_stats.csv python $PWD/profiles/ddp_prof.py
fi
if want pmctrain; then   # counters for config 3 (MFMA busy, FETCH / WRITE per training kernel), fp32 and autocast
  profiles/pmc.sh /tmp/pmc_r06t --mode train > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r06t $o/r06_train_pmc_summary.csv r06train > /dev/null
  rm -rf /tmp/pmc_r06t
  profiles/pmc.sh /tmp/pmc_r06ta --mode train --dtype bf16 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r06ta $o/r06_train_bf16_pmc_summary.csv r06trainbf16 > /dev/null
  rm -rf /tmp/pmc_r06ta
fi
if want bf16; then
  profiles/prof.sh $o/r06_bf16_kernel_stats.csv python $PWD/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline
  profiles/prof.sh $o/r06_split_kernel_stats.csv python $PWD/bench.py --dtype split --steps 5 --warmup 2 --no-cpu-baseline
  profiles/pmc.sh /tmp/pmc_r06b --dtype bf16 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r06b $o/r06_bf16_pmc_summary.csv r06bf16 > /dev/null
  rm -rf /tmp/pmc_r06b
fi
if want pmc; then
  profiles/pmc.sh /tmp/pmc_r06 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r06 $o/r06_pmc_summary.csv r06 > /dev/null
  rm -rf /tmp/pmc_r06
  # counters for config 2 (B = 64) as well
  profiles/pmc.sh /tmp/pmc_r06b --shape config2 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r06b $o/r06_pmc_summary_b64.csv r06b64 > /dev/null
  rm -rf /tmp/pmc_r06b $o/r06b64_trunk_hbm_bytes.json
  profiles/pmc.sh /tmp/pmc_r06s --dtype split > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r06s $o/r06_split_pmc_summary.csv r06split > /dev/null
  rm -rf /tmp/pmc_r06s
fi
if want sweep; then
  python profiles/parity_sweep.py 2>/dev/null | grep '^{' > $o/r06_parity_sweep.jsonl
  python bench.py --shape config2 --no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra 2>/dev/null | grep '^{' > $o/r06_config2_bench.json
fi
