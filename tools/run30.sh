for a in "--streams 4" "--streams 3" "--streams 6" "--streams 8"; do
timeout 600 python bench.py $a --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$a', d['value'], 'ms/step', d['ms_per_step'], 'conc', r['concurrency'], 'kavg', r['kernel_ms_avg'], 'serial', r['serial']['mrays_per_s'], 'frac', r['frac'])"
done
