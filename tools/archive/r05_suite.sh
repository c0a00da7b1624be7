#!/bin/bash
# one full GPU suite + the default bench line (+ N runs of the in-flight stress in fresh processes)
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -rs 2>&1 | tail -15 > gpurun_out/r05_suite.txt
cat gpurun_out/r05_suite.txt
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench rc $?"
tail -c 600 gpurun_out/r05_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05_bench.json') if l.startswith('{')][-1])
pl = d.get('pipeline', {})
print('value %.4g  frac %.3f  ms/step %.4f' % (d['value'], d['roofline']['frac'], d['ms_per_step']))
for k in ('roofline',):
    print('pipeline single frac %.3f ms %.3f' % (pl['roofline']['frac'], pl['ms']))
print('batched', pl['batched']['roofline']['frac'], 'overlapped', pl['overlapped']['roofline']['frac'])
print('sparse', json.dumps({k: pl['sparse'][k] for k in ('records', 'rows', 'ms', 'kmers_per_s')}), pl['sparse']['roofline']['frac'], pl['sparse']['batched']['roofline']['frac'])
print('feed', json.dumps(pl['feed_inclusive'])[:900])
print('small', pl['small']['roofline']['frac'])
PY
