"""Build libmi355env.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gymnasium_amd.csrc.build [--force]

Flags that matter for parity: -ffp-contract=off (hipcc's device default is `fast`, which would fuse a*b+c into FMAs
the CPU reference does not perform) and no -ffast-math.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = {  # translation unit -> the headers it depends on
    "engine.hip": ["envs_classic.h", "sincos_exact.h", "sincos_table.h", "pow_exact.h", "pow_tables.h", "wrappers_internal.h", "mjx_physics.h", "pcg64_dev.h",
                   "mjx_core.h", "mjx_coop.h", "mjx_kernels.h", "ziggurat_tables.h",
                   os.path.join("generated", "mjx_models.h"), os.path.join("..", "..", "include", "mi355env.h")],
    "physics16.hip": ["mjx_physics.h", "envs_classic.h", "sincos_exact.h", "pow_exact.h", "pcg64_dev.h", "mjx_core.h", "mjx_coop.h", "mjx_kernels.h", os.path.join("generated", "mjx_models.h")],
    "physics32.hip": ["mjx_physics.h", "envs_classic.h", "sincos_exact.h", "pow_exact.h", "pcg64_dev.h", "mjx_core.h", "mjx_coop.h", "mjx_kernels.h", os.path.join("generated", "mjx_models.h")],
    "wrappers.hip": ["wrappers_internal.h", os.path.join("..", "..", "include", "mi355env.h")],
    # engine.hip once more with MI_CLASSIC_TU: only the classic-control kernels and their launchers (namespace mi_classic)
    "classic.hip": ["engine.hip", "envs_classic.h", "sincos_exact.h", "sincos_table.h", "pow_exact.h", "pow_tables.h", "wrappers_internal.h", "pcg64_dev.h",
                    os.path.join("..", "..", "include", "mi355env.h")],
}
OUT = os.path.join(HERE, "libmi355env.so")
ARCH = "gfx950"
# Per translation unit, on top of FLAGS: LLVM's iterative GCN scheduler for the cooperative physics kernels.  With ONE wavefront per SIMD it
# hides LDS / VALU latency better than the default max-occupancy scheduler (which trades ILP for an occupancy these kernels cannot have):
# Humanoid-v5 +33 %, Ant-v5 +19 %, results bit-identical to the default scheduler's (scripts/coop_phase_bench.hip: 65536 / 32768 envs x 25
# env-steps, every bit of the state) -- PROVIDED the RK4 stage update stays out of line (mjx_coop.h rk4_stage): inlined, the 16-lane Ant instantiation is
# miscompiled (round 2: wrong results under the iterative scheduler; round 3: a memory fault under the DEFAULT scheduler; stand-alone reproducer and the
# mechanism -- SGPR spills into VGPR lanes -- in scripts/repro/README.md, profiles/r03_rk4_inline.txt).
# engine.hip stays on the default scheduler: built with iterative-maxocc its one-lane Humanoid kernel diverged.
ITERATIVE = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]
# ... and MachineLICM told to sink loop invariants back next to their uses when that avoids a spill: these 9 - 21 k-instruction kernels hoist
# hundreds of literal / address materialisations out of the sub-step loop, whose live ranges then cost more than they save (round 2, Ant: 51 -> 4
# spilled VGPRs, 72 -> 11 scratch instructions, +5.7 % in the stand-alone harness; Humanoid Newton +4 %, PGS +2 %, bit-identical results, A/B on
# one box).
# History of the 16-lane unit: in round 2 its LIBRARY kernels (not the harness's) came out WRONG with this flag -- Ant-v5 produced NaNs, caught by
# tests/test_gpu_scheduler_guard.py and tests/test_gpu_mujoco.py -- so it shipped with MachineLICM switched off instead (NO_MLICM).  The cause was never
# isolated.  After the round-3 rewrites of the forward pass the same flag set builds a physics16.hip that is bit-identical to the
# default-scheduler build on all four 16-lane robots, NaN-free, with 0 spilled VGPRs for the Ant (was 13 - 25) and +1.1 % (round-3 calls 25 / 26,
# scripts/r03/README.md, scripts/r03/guard_variant.py): both units now use ONE flag set.  The guard test stays what decides: if it ever fails again,
# NO_MLICM for physics16.hip is the known-good fallback.
SINK = ["-mllvm", "-sink-insts-to-avoid-spills=true"]
NO_MLICM = ["-mllvm", "-disable-machine-licm"]
# classic.hip (round 4): the five classic-control kinds' kernels under LLVM's max-ILP machine scheduler -- one wavefront per SIMD at 65 536 sub-environments,
# so a dependent instruction's latency is hidden by the wavefront's OWN independent instructions or not at all.  A/B on one box (whole engine.hip
# rebuilt with the flag, scripts/ab_bench.py, profiles/r04_maxilp_classic.txt): CartPole +6.4 %, Pendulum +5.8 %, MountainCarContinuous +2.7 %, Acrobot -0.3 %;
# Taxi -1.3 %, Hopper (one-lane) -2.0 % -- hence only the classic kernels moved.  Results are bit-identical by construction (no re-association; every
# classic parity test is array_equal) and the two-build comparison of tests/test_gpu_scheduler_guard.py covers this unit as well.
# (`iterative-ilp` crashes clang-22 on engine.hip, `iterative-maxocc` stops with "Illegal instruction detected: Operand has incorrect register class".)
MAXILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
TU_FLAGS = {"physics16.hip": ITERATIVE + SINK, "physics32.hip": ITERATIVE + SINK, "classic.hip": MAXILP}
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-missing-braces"]


def generate_models() -> None:
    """Regenerate generated/mjx_models.h from the model descriptions (only rewritten when its content changes)."""
    root = os.path.normpath(os.path.join(HERE, "..", ".."))
    if root not in sys.path:
        sys.path.insert(0, root)
    from gymnasium_amd.envs.mujoco import codegen

    codegen.generate()


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in deps)


OUT_REF = os.path.join(HERE, "libmi355env_ref.so")


def build(force: bool = False, verbose: bool = True, reference_scheduler: bool = False) -> str:
    """Compile every stale translation unit to an object file (hipcc, gfx950 only) and link libmi355env.so.

    ``reference_scheduler=True`` builds libmi355env_ref.so instead: the SAME sources with hipcc's default instruction scheduler in every
    translation unit (no TU_FLAGS).  It is test infrastructure for tests/test_gpu_scheduler_guard.py, which requires the shipped
    (iterative-scheduler) cooperative kernels to be bit-identical to it -- the guard against the miscompile described above."""
    generate_models()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = OUT_REF if reference_scheduler else OUT
    objs, relink = [], force or not os.path.exists(out)
    jobs = []
    for src, headers in SOURCES.items():
        tu_flags = [] if reference_scheduler else TU_FLAGS.get(src, [])
        # translation units whose flags do not differ between the two builds share one object file
        suffix = "_ref.o" if (reference_scheduler and TU_FLAGS.get(src)) else ".o"
        obj = os.path.join(HERE, os.path.splitext(src)[0] + suffix)
        objs.append(obj)
        if force or _stale(obj, [src, "build.py"] + headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *tu_flags, *os.environ.get("MI355ENV_HIPCC_FLAGS", "").split(), "-c", "-o", obj,
                   os.path.join(HERE, src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))  # the translation units compile side by side
            relink = True
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if relink or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    return out


def build_both(verbose: bool = True):
    """libmi355env.so and libmi355env_ref.so with all translation units compiling side by side: the checker's own objects (the units that have
    TU_FLAGS in the product) are started first, then the product is built, then the checker is linked from the shared and its own objects."""
    generate_models()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = []
    for src, headers in SOURCES.items():
        if not TU_FLAGS.get(src):
            continue
        obj = os.path.join(HERE, os.path.splitext(src)[0] + "_ref.o")
        if _stale(obj, [src, "build.py"] + headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *os.environ.get("MI355ENV_HIPCC_FLAGS", "").split(), "-c", "-o", obj, os.path.join(HERE, src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
    out = build(verbose=verbose)
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    return out, build(verbose=verbose, reference_scheduler=True)


VERIFY_CHILD = r"""
import sys, hashlib
sys.path.insert(0, {root!r})
import numpy as np
import gymnasium_amd
h = hashlib.sha256()
for env_id, kw in (("Ant-v5", {{"max_episode_steps": 6}}), ("HalfCheetah-v5", {{}}), ("Humanoid-v5", {{}}), ("CartPole-v1", {{}}), ("Pendulum-v1", {{}})):
    env = gymnasium_amd.make_vec(env_id, num_envs=512, **kw)
    obs, _ = env.reset(seed=3)
    env.action_space.seed(1)
    h.update(obs.tobytes())
    for t in range(24 if env_id == "Humanoid-v5" else 10):
        o, r, te, tr, _ = env.step(env.action_space.sample())
        h.update(o.tobytes()), h.update(r.tobytes()), h.update(te.tobytes()), h.update(tr.tobytes())
    h.update(env.get_state()[0].tobytes())
    env.close()
print("DIGEST", h.hexdigest())
"""


class MiscompileError(RuntimeError):
    """The shipped build and the default-scheduler build of the same sources produced different bits (verify_on_device)."""


class GuardNotRun(RuntimeError):
    """verify_on_device could not run one of its children (GPU busy, out of memory, import error ...): infrastructure, not a verdict."""


def verify_on_device(timeout_s: float = 300.0) -> str:
    """On a GPU box: the shipped cooperative physics kernels (iterative scheduler + MachineLICM settings, TU_FLAGS) against the SAME sources under
    hipcc's defaults (libmi355env_ref.so), every bit of a short trajectory that includes finished episodes and resets, each build in its own child
    process.  The scheduler settings have a history of exposing a code-generation defect (scripts/repro/README.md); this is the check that a new
    toolchain, box or source change did not bring it back, cheap enough to run with every smoke() (tests/test_gpu_scheduler_guard.py is the long
    form).  Returns the common digest; raises MiscompileError if the builds differ and GuardNotRun when a child process failed for any other reason.  (Spilling SGPRs to memory instead -- the documented cure of the defect -- costs
    x0.45 - 0.6 on these kernels: profiles/r04_two_waves.txt.)"""
    root = os.path.normpath(os.path.join(HERE, "..", ".."))
    digests = {}
    for lib in (OUT, OUT_REF):
        if not os.path.exists(lib):
            raise FileNotFoundError(f"{lib} missing: build with `python -m gymnasium_amd.csrc.build --ref`")
        p = subprocess.run([sys.executable, "-c", VERIFY_CHILD.format(root=root)], env=dict(os.environ, MI355ENV_LIBRARY=lib), capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("DIGEST ")]
        if p.returncode != 0 or not lines:
            raise GuardNotRun(f"verify_on_device: {os.path.basename(lib)} failed: {p.stdout[-500:]} {p.stderr[-1500:]}")
        digests[lib] = lines[-1].split()[1]
    if digests[OUT] != digests[OUT_REF]:
        raise MiscompileError("the shipped cooperative physics kernels differ from the default-scheduler build of the same sources (miscompile guard, "
                           f"gymnasium_amd/csrc/build.py TU_FLAGS): {digests}")
    return digests[OUT]


if __name__ == "__main__":
    if "--ref" in sys.argv and "--force" not in sys.argv:
        print(*build_both())
    else:
        print(build(force="--force" in sys.argv))
        if "--ref" in sys.argv:
            print(build(force=True, reference_scheduler=True))
