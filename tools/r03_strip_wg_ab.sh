source tools/gpu_steps.sh
O=gpurun_out/r03
mkdir -p $O
line() { python3 -c "
import json,sys
l = json.loads([x for x in open('$O/probe.json').read().splitlines() if x.startswith('{')][-1]); r = l['roofline']
print('$1: value %.0f GCUPS ms/step %.3f kernel_ms %.3f frac %s %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('frac'), l.get('invalid','')))"; }
for form in wg4 wg1 wg4 wg1; do
  if [ $form = wg1 ]; then export PWA_STRIP_WG1=1; else unset PWA_STRIP_WG1; fi
  step c3 200 python3 bench.py --no-cpu-baseline --steps 5 > $O/probe.json 2>/dev/null; line "c3 $form"
  step c3i 200 python3 bench.py --workload c3i --no-cpu-baseline --steps 10 > $O/probe.json 2>/dev/null; line "c3i $form"
  step c4 200 python3 bench.py --workload c4 --no-cpu-baseline --steps 5 > $O/probe.json 2>/dev/null; line "c4 $form"
done
unset PWA_STRIP_WG1
