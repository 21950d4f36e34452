for f in "" "--no-fuse"; do for st in "" "--streams 1" "--chunk 9"; do
python bench.py --no-cpu-baseline --steps 12 $f $st 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['regions'];print('$f $st', 'decode',r['decode']['ms_per_step'],'encode',r['encode']['ms_per_step'],'e2e',r['encode_decode_score']['ms_per_step'],'w1',r['w1_encode_decode_score']['ms_per_step'])"
done; done
