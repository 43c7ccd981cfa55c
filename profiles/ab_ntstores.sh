#!/bin/bash
# A/B on ONE box: the shipped library (large activation saves as non-temporal stores, csrc/catre_device.h st_stream) against
# the same sources built with -DCATRE_NT_SAVES=0 (catre_amd/csrc/libcatre_hip_plainst.so).
#   profiles/ab_ntstores.sh > gpurun_out/r04_ab_ntstores.jsonl
for rep in 1 2; do
for lib in "" "$PWD/catre_amd/csrc/libcatre_hip_plainst.so"; do
  tag=$([ -z "$lib" ] && echo nt || echo plain)
  for args in "--steps 10 --warmup 3 --no-split-extra" "--dtype split --steps 10 --warmup 3" "--dtype bf16 --steps 10 --warmup 3" \
              "--mode train --steps 4 --warmup 3" "--mode train --dtype bf16 --steps 4 --warmup 3" "--mode train --dtype split --steps 4 --warmup 3"; do
    CATRE_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-train-extra --no-small-extra 2>/dev/null | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'build':'$tag','args':'$args','value':d['value'],'ms_per_step':d['ms_per_step'],'roofline':d.get('roofline',{}).get('frac')}))"
  done
done
done
