"""CPU checks of the boundary: the library loads, exports every symbol include/mi355env.h declares, refuses to run
without a GPU (no CPU fallback), and the ctypes structs match the header's layout."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "mi355env.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gymnasium_amd import _native

    lib = _native.load_library()
    declared = header_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib.dll, name), f"{name} declared in include/mi355env.h but not exported"
    assert sorted("mi_" + s for s in _native.SYMBOLS + _native.WRAPPER_SYMBOLS + _native.HOST_SYMBOLS) == declared
    assert lib.abi_version() == _native.ABI_VERSION


def test_oracle_exports_the_same_abi(oracle_factory):
    from gymnasium_amd import _native
    from oracle import oracle

    lib = oracle.load()
    for s in _native.SYMBOLS:
        assert hasattr(lib.dll, "orc_" + s)


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors == what a C compiler makes of include/mi355env.h: sizes, and the offset of every field (gcc is in the image)."""
    import shutil
    import subprocess

    from gymnasium_amd import _native as n

    assert ctypes.sizeof(n.MiConfig) == 8 * 4 + 16 * 8
    assert ctypes.sizeof(n.MiLayout) == 8 * 4
    assert ctypes.sizeof(n.MiStepIO) == 11 * 8 + 2 * 4  # ten pointers, actions_dtype, reserved, actions_out (ABI 7)
    assert ctypes.sizeof(n.MiRolloutIO) == 6 * 8 + 2 * 4
    assert ctypes.sizeof(n.MiStats) == 5 * 8
    assert (n.MI_F32, n.MI_F64, n.MI_I64, n.MI_F64_WEAK) == (0, 1, 2, 3)
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler to cross-check the offsets with")
    pairs = {"mi_config": n.MiConfig, "mi_layout": n.MiLayout, "mi_step_io": n.MiStepIO, "mi_rollout_io": n.MiRolloutIO, "mi_stats": n.MiStats,
             "mi_tabular_table": n.MiTabularTable, "mi_step_epilogue": n.MiStepEpilogue}
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{os.path.join(ROOT, "include", "mi355env.h")}"', "int main(void) {"]
    for cname, ct in pairs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append(f'printf("abi %d\\n", MI355ENV_ABI_VERSION); return 0; }}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-o", str(tmp_path / "layout"), str(src)], check=True)
    out = dict(line.split() for line in subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in pairs.items():
        assert int(out[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"
    assert int(out["abi"]) == n.ABI_VERSION


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly; it must never route through the oracle."""
    import gymnasium_amd
    from gymnasium_amd import _native

    lib = _native.load_library()
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.NativeError) as e:
        gymnasium_amd.make_vec("CartPole-v1", num_envs=4)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    # the package never imports the oracle
    import sys

    assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules if "gymnasium_amd" in getattr(sys.modules[m], "__name__", "") )
    src = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gymnasium_amd")):
        src += [os.path.join(dirpath, f) for f in files if f.endswith((".py", ".hip", ".h", ".cpp"))]
    for f in src:
        text = open(f).read()
        assert "liboracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_bench_knows_which_envs_run_the_two_role_rollout():
    """bench.py names the dominant kernel for its rocprofv3 counter passes: its table must follow the traits in envs_classic.h."""
    import re

    import bench

    src = open(os.path.join(ROOT, "gymnasium_amd", "csrc", "envs_classic.h")).read()
    traits = {}
    for m in re.finditer(r"struct (\w+)T \{(.*?)\n\};", src, re.S):
        body = m.group(2)
        duo = re.search(r"static constexpr bool DUO_ROLLOUT = (true|false)", body)
        chunk = re.search(r"static constexpr int DUO_CHUNK = (\d+)", body)
        if duo:
            traits[m.group(1)] = int(chunk.group(1)) if duo.group(1) == "true" else None
    ids = {"CartPole": "CartPole-v1", "Pendulum": "Pendulum-v1", "Acrobot": "Acrobot-v1", "MountainCar": "MountainCar-v0", "MountainCarContinuous": "MountainCarContinuous-v0"}
    assert set(traits) == set(ids), traits
    assert {ids[k]: v for k, v in traits.items() if v} == bench.DUO_CHUNK
