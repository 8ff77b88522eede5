#!/usr/bin/env python3
"""A/B throughput of library builds on ONE GPU box: every (library, env) pair runs `bench.py` in its own process, interleaved A B A B.

    python scripts/ab_bench.py --libs product=gymnasium_amd/csrc/libmi355env.so w2=gymnasium_amd/csrc/libmi355env_w2.so \
        --envs Hopper-v5:65536:4 Walker2d-v5:65536:4 [--rounds 2] [--coop] [--env-kwargs '{}'] [--out gpurun_out/ab.txt]

Prints one line per (env, library): env-steps/s of every round and their mean, avg kernel ms.  Experiment infrastructure only."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True, help="name=path pairs (name=path@KEY=VAL,... adds environment variables)")
    ap.add_argument("--envs", nargs="+", required=True, help="env_id:num_envs:inner")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--coop", action="store_true", help="MI355ENV_MJ_COOP=1: the cooperative physics kernel also where the one-lane kernel ships")
    ap.add_argument("--serial", action="store_true", help="MI355ENV_MJ_SERIAL=1")
    ap.add_argument("--env-kwargs", default="{}")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--warmup", type=int, default=None, help="untimed launches before the timed region (e.g. 40 with terminate_when_unhealthy=false: robots on the ground)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    libs = [x.split("=", 1) for x in args.libs]
    res = {}
    for rnd in range(args.rounds):
        for spec in args.envs:
            env_id, n, inner = spec.split(":")
            for name, path in libs:
                path, _, extra = path.partition("@")  # name=path@KEY=VAL,KEY=VAL: the same library under other environment switches
                env = dict(os.environ, MI355ENV_LIBRARY=os.path.abspath(os.path.join(ROOT, path)))
                env.update(kv.split("=", 1) for kv in extra.split(",") if kv)
                if args.coop:
                    env["MI355ENV_MJ_COOP"] = "1"
                if args.serial:
                    env["MI355ENV_MJ_SERIAL"] = "1"
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--env", env_id, "--num-envs", n, "--inner", inner, "--no-api", "--no-cpu-baseline",
                       "--no-secondary", "--pmc", "off", "--sustained", "0", "--pilot-seconds", str(args.seconds), "--env-kwargs", args.env_kwargs]
                if args.warmup is not None:
                    cmd += ["--warmup", str(args.warmup)]
                p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
                try:
                    r = json.loads(p.stdout.strip().splitlines()[-1])
                    res.setdefault((spec, name), []).append((r["value"], r["roofline"]["avg_kernel_ms"]))
                except Exception:
                    res.setdefault((spec, name), []).append((float("nan"), float("nan")))
                    print("FAILED", spec, name, p.stderr[-600:], flush=True)
    lines = []
    for spec in args.envs:
        base = None
        for name, _ in libs:
            vals = res[(spec, name)]
            mean = sum(v for v, _ in vals) / len(vals)
            base = base or mean
            lines.append(f"{spec:34s} {name:12s} " + " ".join(f"{v / 1e6:9.3f}" for v, _ in vals) + f"  mean {mean / 1e6:9.3f} M env-steps/s ({mean / base:5.3f}x)  launch {vals[-1][1]:8.3f} ms")
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "a") as f:
            f.write("# " + " ".join(sys.argv[1:]) + "\n" + text + "\n")


if __name__ == "__main__":
    main()
