#!/bin/bash
# A/B of an environment switch on one box: alternates `bench.py` runs with and without it (box-to-box variance is +-3 %, so both
# legs must come from the same call).   usage: tools/ab_env.sh "VAR=1 [VAR2=..]" [rounds] [extra bench flags]
sw="$1"; rounds="${2:-2}"; shift; shift
get() { python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"; }
for i in $(seq 1 "$rounds"); do
  echo "off  $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | get)"
  echo "on   $(env $sw python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | get)   [$sw]"
done
