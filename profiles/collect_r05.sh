#!/bin/bash
# Everything the round-5 docs quote, collected on a GPU box from the build of this commit (run from the repo root), in
# parts that each finish within a few minutes and write only small summaries:
#   profiles/collect_r05.sh tests|bench|small|prof|train|pmc|pmctrain|bf16|sweep      -> gpurun_out/r05_*   (copy the summaries into profiles/)
o=gpurun_out
part=${1:-all}
want() { [ "$part" = all ] || [ "$part" = "$1" ]; }
if want tests; then
  python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^[0-9]+ passed|^[0-9]+ failed| passed| failed|^ERROR|^FAILED" | tail -8 > $o/r05_gputest.log
  cat $o/r05_gputest.log
fi
if want bench; then
  python bench.py > $o/r05_bench_n1.json 2> $o/r05_bench_n1.err
  { python bench.py --dtype split --no-cpu-baseline; python bench.py --dtype bf16 --no-cpu-baseline;
    python bench.py --shape config5 --no-cpu-baseline --steps 5; python bench.py --shape config5 --dtype bf16 --no-cpu-baseline --steps 5;
    python bench.py --shape config5 --dtype split --no-cpu-baseline --steps 5; } 2>/dev/null | grep '^{' > $o/r05_modes_bench.jsonl
  { python bench.py --mode train --steps 4 --warmup 4; python bench.py --mode train --dtype bf16 --steps 4 --warmup 4;
    python bench.py --mode train --dtype split --steps 4 --warmup 4; } 2>/dev/null | grep '^{' > $o/r05_train_bench.jsonl
fi
if want small; then
  python profiles/small_batch.py 2>/dev/null | grep '^{' > $o/r05_other_configs.jsonl
  python profiles/small_sweep.py 1 2 3 4 6 8 16 32 2>/dev/null | grep '^{' > $o/r05_small_sweep.jsonl
  python profiles/eval_loop_probe.py 1 2 4 6 2>/dev/null | grep '^{' > $o/r05_eval_loop.jsonl
  { python profiles/multi_stream_probe.py 1 1 2 4 8; python profiles/multi_stream_probe.py 4 1 2 4; } 2>/dev/null | grep '^{' > $o/r05_multi_stream.jsonl
  profiles/prof.sh $o/r05_b1_kernel_stats.csv python $PWD/profiles/b1_profile.py 1 50
fi
if want prof; then
  profiles/prof.sh $o/r05_kernel_stats.csv python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra
fi
if want train; then
  python profiles/train_step.py 16 64 256 2>/dev/null | grep '^{' > $o/r05_train_step.jsonl
  PROF_TRACE="$PWD/$o/r05_train_trace.csv 1500" profiles/prof.sh $o/r05_train_kernel_stats.csv python $PWD/bench.py --mode train --steps 3 --warmup 1
  python profiles/gemm_probe.py 2>/dev/null | grep '^{' > $o/r05_gemm_probe.jsonl
  profiles/prof.sh $o/r05_train_bf16_kernel_stats.csv python $PWD/bench.py --mode train --dtype bf16 --steps 3 --warmup 1
  profiles/prof.sh $o/r05_train_split_kernel_stats.csv python $PWD/bench.py --mode train --dtype split --steps 3 --warmup 1
fi
if want pmctrain; then   # counters for config 3 (MFMA busy, FETCH / WRITE per training kernel)
  profiles/pmc.sh /tmp/pmc_r05t --mode train > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r05t $o/r05_train_pmc_summary.csv r05train > /dev/null
  rm -rf /tmp/pmc_r05t
fi
if want bf16; then
  profiles/prof.sh $o/r05_bf16_kernel_stats.csv python $PWD/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline
  profiles/prof.sh $o/r05_split_kernel_stats.csv python $PWD/bench.py --dtype split --steps 5 --warmup 2 --no-cpu-baseline
  profiles/pmc.sh /tmp/pmc_r05b --dtype bf16 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r05b $o/r05_bf16_pmc_summary.csv r05bf16 > /dev/null
  rm -rf /tmp/pmc_r05b
fi
if want pmc; then
  profiles/pmc.sh /tmp/pmc_r05 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r05 $o/r05_pmc_summary.csv r05 > /dev/null
  rm -rf /tmp/pmc_r05
  # counters for config 2 (B = 64) as well
  profiles/pmc.sh /tmp/pmc_r05b --shape config2 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r05b $o/r05_pmc_summary_b64.csv r05b64 > /dev/null
  rm -rf /tmp/pmc_r05b $o/r05b64_trunk_hbm_bytes.json
  profiles/pmc.sh /tmp/pmc_r05s --dtype split > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r05s $o/r05_split_pmc_summary.csv r05split > /dev/null
  rm -rf /tmp/pmc_r05s
fi
if want sweep; then
  python profiles/parity_sweep.py 2>/dev/null | grep '^{' > $o/r05_parity_sweep.jsonl
  python bench.py --shape config2 --no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra 2>/dev/null | grep '^{' > $o/r05_config2_bench.json
fi
