#!/bin/bash
# Everything the round-2 docs quote, collected on a GPU box from the build of this commit (run from the repo root), in
# parts that each finish within a few minutes and write only small summaries:
#   profiles/collect_r02.sh tests|bench|small|prof|train|pmc      -> gpurun_out/r02_*   (copy the summaries into profiles/)
o=gpurun_out
part=${1:-all}
want() { [ "$part" = all ] || [ "$part" = "$1" ]; }
if want tests; then
  python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -5 > $o/r02_gputest.log
  cat $o/r02_gputest.log
fi
if want bench; then
  python bench.py > $o/r02_bench_n1.json 2> $o/r02_bench_n1.err
  { python bench.py --dtype split --no-cpu-baseline; python bench.py --dtype bf16 --no-cpu-baseline;
    python bench.py --shape config5 --no-cpu-baseline --steps 5; python bench.py --shape config5 --dtype bf16 --no-cpu-baseline --steps 5;
    python bench.py --shape config5 --dtype split --no-cpu-baseline --steps 5; } 2>/dev/null | grep '^{' > $o/r02_modes_bench.jsonl
  { python bench.py --mode train --steps 4 --warmup 4; python bench.py --mode train --dtype bf16 --steps 4 --warmup 4;
    python bench.py --mode train --dtype split --steps 4 --warmup 4; } 2>/dev/null | grep '^{' > $o/r02_train_bench.jsonl
fi
if want small; then
  python profiles/small_batch.py 2>/dev/null | grep '^{' > $o/r02_other_configs.jsonl
  python profiles/small_sweep.py 1 2 3 4 6 8 16 32 2>/dev/null | grep '^{' > $o/r02_small_sweep.jsonl
  python profiles/eval_loop_probe.py 1 2 4 6 2>/dev/null | grep '^{' > $o/r02_eval_loop.jsonl
  { python profiles/multi_stream_probe.py 1 1 2 4 8; python profiles/multi_stream_probe.py 4 1 2 4; } 2>/dev/null | grep '^{' > $o/r02_multi_stream.jsonl
  profiles/prof.sh $o/r02_b1_kernel_stats.csv python $PWD/profiles/b1_profile.py 1 50
fi
if want prof; then
  profiles/prof.sh $o/r02_kernel_stats.csv python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra --no-small-extra
fi
if want train; then
  python profiles/train_step.py 16 64 256 2>/dev/null | grep '^{' > $o/r02_train_step.jsonl
  PROF_TRACE="$PWD/$o/r02_train_trace.csv 1500" profiles/prof.sh $o/r02_train_kernel_stats.csv python $PWD/bench.py --mode train --steps 3 --warmup 1
  python profiles/gemm_probe.py 2>/dev/null | grep '^{' > $o/r02_gemm_probe.jsonl
fi
if want pmc; then
  profiles/pmc.sh /tmp/pmc_r02 > /dev/null 2>&1
  python profiles/summarize_pmc.py /tmp/pmc_r02 $o/r02_pmc_summary.csv r02 > /dev/null
  rm -rf /tmp/pmc_r02
fi
