cd /root/repo
for n in 786432 983040 1000000 1048576 1179648; do
python bench.py --ns-per-gpu $n --no-dense --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, 'ms', round(d['ms_per_step'],2), 'steps/s %.4e'%d['value'], 'ns/ms', round($n/d['ms_per_step'],1))"
done
