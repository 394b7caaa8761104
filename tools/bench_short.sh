for i in 1 2 3; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20-step: %.3f G ms/step %.4f kernel %.4f resets %s %s' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['config']['resets_in_window'], d['config']['timed_as']))"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20-step eager: %.3f G ms/step %.4f kernel %.4f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
python3 bench.py --no-cpu-baseline --no-fused 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('500-step: %.3f G ms/step %.4f kernel %.4f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
