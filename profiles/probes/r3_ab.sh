#!/bin/bash
# usage: r3_ab.sh <outfile> <dtype> <tag> [VAR=VAL ...]   — one short bench run, appends "tag dtype img/s ms" to outfile
out=$1; dt=$2; tag=$3; shift 3
env "$@" python bench.py --steps 20 --warmup 5 --dtype $dt --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "$tag $dt FAILED" >> $out; tail -3 gpurun_out/_ab.err >> $out; exit 0; }
python - "$out" "$tag" "$dt" <<PY
import json,sys
d=json.load(open('gpurun_out/_ab.json'))
open(sys.argv[1],'a').write('%s %s %.1f %.3f\n'%(sys.argv[2],sys.argv[3],d['value'],d['ms_per_step']))
PY
