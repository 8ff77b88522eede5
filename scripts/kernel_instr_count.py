"""Static instruction counts per kernel of a gfx950 object / library: total, and the largest loop body (backward branch span), by class.
usage: python scripts/kernel_instr_count.py [file] [regex over demangled kernel names]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import ROOT, extract_all

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def classify(op):
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith(("v_", "ds_swizzle")):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gymnasium_amd", "csrc", "libmi355env.so")
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for co in extract_all(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co), f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        kernels, cur = {}, None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                kernels[cur] = []
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", line)
            if m and cur:
                kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
        names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
        for (mangled, ins), name in zip(kernels.items(), names):
            if pat and not pat.search(name) or not ins:
                continue
            addr = {a: i for i, (a, _, _) in enumerate(ins)}
            # largest backward branch span = the main loop
            best = (0, 0, 0)
            for i, (a, op, args) in enumerate(ins):
                if op.startswith(("s_cbranch", "s_branch")):
                    m = re.search(r"(-?\d+)", args)
                    if m:
                        off = int(m.group(1))
                        if off > 32767:
                            off -= 65536
                        tgt = a + 4 + 4 * off
                        if tgt in addr and addr[tgt] < i and i - addr[tgt] > best[0]:
                            best = (i - addr[tgt], addr[tgt], i)
            def summary(sub):
                c = {}
                for _, op, _ in sub:
                    c[classify(op)] = c.get(classify(op), 0) + 1
                return " ".join(f"{k}={v}" for k, v in sorted(c.items()))
            print(name[:110])
            print(f"   total {len(ins)}: {summary(ins)}")
            if best[0]:
                print(f"   largest loop {best[0] + 1}: {summary(ins[best[1]:best[2] + 1])}")


if __name__ == "__main__":
    main()
