# A/B of the two-role rollout kernel against the one-role kernel through bench.py (HIP-event timing): bash scripts/r04/duo_ab_bench.sh [env ids...]
for env in ${@:-CartPole-v1 Pendulum-v1 Acrobot-v1 MountainCarContinuous-v0 MountainCar-v0}; do
  for duo in 0 1 0 1; do
    MI355ENV_ROLLOUT_DUO=$duo python bench.py --env $env --no-secondary --no-api --no-cpu-baseline --pmc off 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env duo=$duo value %.4g kernel_ms %.5f frac %.3f' % (r['value'], r['roofline']['avg_kernel_ms'], r['roofline']['frac']))"
  done
done
