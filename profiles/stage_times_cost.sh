for v in 0 1; do
 if [ $v = 1 ]; then export SUBZERO_NO_STAGE_TIMES=1; fi
 for rep in 1 2 3; do python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-clip-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('no_times=$v ms/step %.4f device %.4f' % (d['ms_per_step'], d['device_ms_per_step']))"; done
done
