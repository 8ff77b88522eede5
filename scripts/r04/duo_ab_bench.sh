# A/B of the multi-role rollout kernels against the one-role kernel through bench.py (HIP-event timing): bash scripts/r04/duo_ab_bench.sh [env ids...]
# MI355ENV_ROLLOUT_DUO: 0 one role, 1 (default) two roles, 3 three roles
for env in ${@:-CartPole-v1 Pendulum-v1 Acrobot-v1 MountainCarContinuous-v0 MountainCar-v0}; do
  for duo in ${DUO_VARIANTS:-0 1 3 0 1 3}; do
    MI355ENV_ROLLOUT_DUO=$duo python bench.py --env $env --no-secondary --no-api --no-cpu-baseline --pmc off 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env roles=$duo value %.4g kernel_ms %.5f frac %.3f' % (r['value'], r['roofline']['avg_kernel_ms'], r['roofline']['frac']))"
  done
done
