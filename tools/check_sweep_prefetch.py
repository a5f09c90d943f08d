#!/usr/bin/env python3
"""Disassembly check of the L2 sweep's descriptor prefetch (fdcm_sweep.hip local_run).

local_run issues the next column's `s_load_dwordx4` from inline assembly WITHOUT a wait: the data lands behind the pop
loop's own `s_waitcnt lgkmcnt(0)`.  The compiler does not know that those four SGPRs are in flight, so nothing in the
language stops it from copying or spilling them before the wait.  This script reads the code the compiler actually
produced: in every kernel of libfdcm_hip.so that contains such a load, from an `s_load_dwordx4 s[a:b]` (other than the
compiler's own loads of the kernel arguments, base s[0:1]) that is not
followed by its own wait up to the next `s_waitcnt` that covers lgkmcnt(0), no instruction may name a register of
s[a:b].  Exit 0 and one line per kernel when that holds; non-zero with the offending lines otherwise.

Usage: check_sweep_prefetch.py [path/to/libfdcm_hip.so]   (needs /opt/rocm/lib/llvm/bin; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, a in enumerate(starts):
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, f"co{i}.o")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if r.returncode == 0 and os.path.getsize(co) > 0:
            out.append(co)
    return out


def sregs(text):
    """SGPR numbers an operand string names: s7, s[8:11]."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return regs


def covers_lgkm0(ins):
    return ins.startswith("s_waitcnt") and (re.search(r"lgkmcnt\(0\)", ins) is not None or re.fullmatch(r"s_waitcnt\s+0(x0+)?", ins.strip()) is not None)


def check(so):
    bad, seen = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            kernel, lines = None, []
            for ln in dis.splitlines() + ["0000 <end>:"]:
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
                if m:
                    if kernel and "k_sweep" in kernel:
                        n, b = scan(kernel, lines)
                        seen += n
                        bad += b
                        if n:
                            print(f"{kernel}: {n} descriptor prefetch(es) in flight, " + ("VIOLATED" if b else "untouched until the wait"))
                    kernel, lines = m.group(1), []
                elif kernel:
                    ins = ln.split("//")[0].strip()
                    if ins:
                        lines.append(ins)
    return seen, bad


def scan(kernel, lines):
    n, bad = 0, []
    for i, ins in enumerate(lines):
        if not ins.startswith("s_load_dwordx4"):
            continue
        ops = [o.strip() for o in ins[len("s_load_dwordx4"):].split(",")]
        if len(ops) >= 2 and ops[1] == "s[0:1]":
            continue  # the compiler's own loads of the kernel arguments (base = the kernarg pointer): it tracks those itself
        dst = sregs(ins.split(",")[0])
        if i + 1 < len(lines) and covers_lgkm0(lines[i + 1]):
            continue  # waited for on the spot
        n += 1
        for j in range(i + 1, len(lines)):
            nxt = lines[j]
            if covers_lgkm0(nxt):
                break
            if nxt.startswith(("s_endpgm", "s_setpc")):
                bad.append(f"{kernel}: `{ins}` never waited for")
                break
            if nxt.startswith(("s_cbranch", "s_branch")):
                continue  # (the pop loop's own branches; the registers are checked on the fall-through text, which holds the whole loop)
            if sregs(nxt) & dst:
                bad.append(f"{kernel}: `{nxt}` touches the destination of `{ins}` before a wait (line +{j - i})")
                break
    return n, bad


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "openfdcm_amd", "libfdcm_hip.so")
    seen, bad = check(so)
    for b in bad:
        print("VIOLATION:", b)
    if seen == 0:
        print("no in-flight s_load_dwordx4 found in any k_sweep kernel: the check does not see what it is for")
        sys.exit(2)
    print("sweep prefetch check", "FAILED" if bad else "ok")
    sys.exit(1 if bad else 0)
