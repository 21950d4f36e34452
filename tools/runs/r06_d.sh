cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_d.json 2> gpurun_out/r06_bench_d.err; echo "bench.py wall: ${SECONDS} s"; tail -2 gpurun_out/r06_bench_d.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_d.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['sustained']['value'], d['config']['launch_schedule'][:260])
r=d['regions']
for k in ('decode','encode','encode_decode_score','w1_encode_decode_score','decompress','compress'):
    print(k, r[k]['ms_per_step'], (r[k].get('roofline') or {}).get('frac_of_fp32_mfma_peak'))
print(r['evaluate_b1'])
PY
