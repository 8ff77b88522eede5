"""Static guards on the built library's ISA (llvm-objdump of libmi355env.so; no GPU needed).

1. Inline-asm Horner steps (sincos_exact.h / pow_exact.h `fma_k<true>`) only in kernels that stay inside the VGPR file.  Inline asm is
   opaque to the compiler's hazard recognizer; in round 3 the one classic kernel whose live values overflow into AGPRs -- Acrobot's fused
   rollout -- gave results that were not reproducible from launch to launch with the asm form (and were with the builtin).  Acrobot is
   therefore instantiated with ExactMathT<false>; every kernel instantiated with the asm-carrying policies must not touch AGPRs at all.
2. The classic-control and ToyText kernels do not spill to scratch.
3. The cooperative MuJoCo kernels keep the resources their throughput rests on: at most a quarter of a CU's LDS per workgroup (four wavefronts per CU,
   one per SIMD -- one byte more and it is three), and no spilled vector registers in the shipped Ant / Humanoid instantiations.
"""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "gymnasium_amd", "csrc", "libmi355env.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(LIB)), reason="needs the ROCm llvm-objdump and the built library")


@pytest.fixture(scope="module")
def kernels():
    """{demangled kernel name: [mnemonic, ...]} of every gfx950 kernel in the library's engine code object(s)."""
    from kernel_resources import extract_all

    out = {}
    for co in extract_all(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co), f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur, per = None, {}
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                per[cur] = []
                continue
            m = re.match(r"^\s+(\S+)", line)
            if m and cur:
                per[cur].append(m.group(1))
        names = subprocess.run(["c++filt"], input="\n".join(per), capture_output=True, text=True).stdout.splitlines()
        for (_, ins), name in zip(per.items(), names):
            if ins:
                out[name] = ins
    return out


def test_inline_asm_fma_only_in_kernels_without_agpr_traffic(kernels):
    asm_policies = ("ExactMathT<true>", "CartPoleT<mi::FastMath>")  # the instantiations whose trig / pow routines carry fma_k<true>
    checked = 0
    for name, ins in kernels.items():
        if any(p in name for p in asm_policies):
            checked += 1
            n = sum(op.startswith("v_accvgpr") for op in ins)
            assert n == 0, f"{name[:140]}: {n} AGPR moves in a kernel with inline-asm v_fma_f64 (instantiate it with ExactMathT<false>)"
    assert checked >= 20, f"only {checked} kernels matched: the policy names changed?"
    acro = [n for n in kernels if "AcrobotT<" in n and "FastMath" not in n]
    assert acro and all("ExactMathT<false>" in n for n in acro), "Acrobot's exact kernels must use the builtin-fma policy"


def test_classic_and_toytext_kernels_do_not_use_scratch(kernels):
    for name, ins in kernels.items():
        if ("mi::" in name and "T<mi::" in name) or "tab_" in name or "bj_rollout" in name:
            n = sum(op.startswith("scratch_") for op in ins)
            assert n == 0, f"{name[:140]}: {n} scratch instructions"


def test_cooperative_mujoco_kernels_keep_their_lds_and_register_budget():
    from kernel_resources import resources

    rows = [r for r in resources(LIB) if "mj_physics_kernel" in r["name"]]
    assert len(rows) >= 16, [r["name"] for r in rows]
    for r in rows:
        # 160 KB of LDS per CU (MI355X_MICROARCH.md): four workgroups of one wavefront each must fit, or a SIMD stays empty
        assert r["group_segment_fixed_size"] * 4 <= 160 * 1024, f"{r['name']}: {r['group_segment_fixed_size']} B of LDS per workgroup: only three workgroups per CU"
        assert r["vgpr_count"] <= 512
    for robot in ("AntModel", "HumanoidModel", "HumanoidStandupModel"):
        for r in rows:
            if robot + "," in r["name"]:
                assert r.get("vgpr_spill_count", 0) == 0, f"{r['name']}: {r['vgpr_spill_count']} spilled VGPRs (build.py TU_FLAGS, DESIGN.md section 7)"


def test_two_role_rollout_kernels_fit_two_wavefronts_per_simd():
    """rollout_duo_kernel runs an env and an aux wavefront of the same 64 sub-environments on one SIMD (engine.hip): a workgroup is 8 wavefronts on the 4
    SIMDs of a CU, so each wavefront may use at most half the register file, nothing may spill, and the rings must fit the CU's LDS."""
    from kernel_resources import resources

    rows = [r for r in resources(LIB) if "rollout_duo_kernel" in r["name"]]
    assert len(rows) == 8, [r["name"] for r in rows]  # CartPole, Pendulum (round 6: the reward on the aux role), MountainCar, MountainCarContinuous x (exact, fast math)
    assert not any("AcrobotT" in r["name"] or "ActF64" in r["name"] for r in rows), "Acrobot and the float64-row instantiations keep the one-role kernel (envs_classic.h DUO_ROLLOUT)"
    for r in rows:
        assert r["vgpr_count"] + r.get("agpr_count", 0) <= 256, f"{r['name']}: {r['vgpr_count']} registers: two wavefronts no longer fit a SIMD"
        assert r.get("vgpr_spill_count", 0) == 0 and r.get("private_segment_fixed_size", 0) == 0, r["name"]
        assert r["group_segment_fixed_size"] <= 160 * 1024, r["name"]


def test_branch_free_tabular_rollouts_are_instantiated_and_stay_in_registers(kernels):
    """tab_rollout_lean_kernel<KL, FULL, ONE_START, APOW2> (16 instantiations) and bj_rollout_lean_kernel<FULL> (2): what mi_rollout launches for the collector's
    ToyText configurations (engine.hip).  None may spill, and all stay within 128 VGPRs (four wavefronts per SIMD at the batch sizes beyond the benchmark's)."""
    from kernel_resources import resources

    lean = [n for n in kernels if "tab_rollout_lean_kernel<" in n]
    bj = [n for n in kernels if "bj_rollout_lean_kernel<" in n]
    assert len(lean) == 16 and len(bj) == 2, (len(lean), len(bj))
    for r in resources(LIB):
        if "rollout_lean_kernel" in r["name"]:
            assert r.get("vgpr_spill_count", 0) == 0 and r.get("private_segment_fixed_size", 0) == 0, r["name"]
            assert r["vgpr_count"] <= 128, f"{r['name']}: {r['vgpr_count']} VGPRs"


def test_acrobot_rollout_kernels_fit_two_wavefronts_per_simd_and_keep_their_constants_in_registers(kernels):
    """Round 6: the exact-math Acrobot rollouts hold at most 256 registers (beyond 65 536 sub-environments two wavefronts then share a SIMD: x1.4 at 262 144,
    profiles/r06_acrobot_shared_divisor_ab.txt), and the trig routines' float64 constants come from LDS once instead of being re-materialised as literals at every
    use (profiles/r06_acrobot_hot_constants_ab.txt): scalar moves were 12 - 15 % of the kernels' instructions before, 6 % after."""
    from kernel_resources import resources

    rows = [r for r in resources(LIB) if "rollout_kernel<mi::AcrobotT<mi::ExactMathT<false>" in r["name"].replace("> >", ">>").replace(" >", ">")]
    assert len(rows) >= 5, [r["name"] for r in rows]
    for r in rows:
        assert r["vgpr_count"] + r.get("agpr_count", 0) <= 256, f"{r['name']}: {r['vgpr_count']} + {r.get('agpr_count', 0)} registers"
        assert r.get("vgpr_spill_count", 0) == 0 and r.get("private_segment_fixed_size", 0) == 0, r["name"]
    body = [ins for name, ins in kernels.items() if "rollout_kernel<mi::AcrobotT<mi::ExactMathT<false>" in name.replace("> >", ">>").replace(" >", ">")]
    assert len(body) >= 5
    for ins in body:
        moves = sum(1 for op in ins if op in ("s_mov_b32", "s_brev_b32", "s_mov_b64", "s_movk_i32"))
        assert moves <= 0.09 * len(ins), f"{moves} scalar moves in {len(ins)} instructions (12 - 15 % before the constants came from LDS: sincos_exact.h RedK / PolyK, envs_classic.h HOT)"
