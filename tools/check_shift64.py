"""Build check: no 64-bit vector shift of a gfx950 kernel takes its shift amount from the LAST vector register the kernel allocates.

Why: round 3 traced a wrong centroid (sum_x read as 1.0 in lanes 0..5 of ~1 % of the waves, never the same ones) to
    v_lshlrev_b64 v[38:39], v47, v[38:39]          in a kernel that allocates v0..v47
-- the compiler's expansion of (float)(uint64).  That is the erratum LLVM calls Shift64HighRegBug (a 64-bit shift whose amount sits in
the last VGPR of an 8-register block, with the next register not allocated, reads a wrong amount); LLVM works around it for gfx90a
only (GCNHazardRecognizer::fixShift64HighRegBug), this ROCm's gfx950 code generator does not, and the MI355X showed it.
The check disassembles the device code of every object / library given and fails on any such instruction.

    python tools/check_shift64.py obia_amd/csrc/libobia_hip.so [more .o / .so files]
"""
import os, re, shutil, subprocess, sys, tempfile


def _llvm_bin():
    """Directory of llvm-objdump: $ROCM_PATH/lib/llvm/bin, /opt/rocm/lib/llvm/bin, or wherever PATH has it."""
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(root, "lib", "llvm", "bin")
    w = shutil.which("llvm-objdump")
    if w:
        return os.path.dirname(w)
    sys.exit("check_shift64: llvm-objdump not found (looked in $ROCM_PATH/lib/llvm/bin, /opt/rocm/lib/llvm/bin and PATH)")


LLVM = _llvm_bin()
SHIFT = re.compile(r"\b(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\b")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def device_objects(path, tmp):
    local = os.path.join(tmp, os.path.basename(path))   # (llvm-objdump writes the bundles beside its input)
    shutil.copy(path, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "amdgcn" in f]


def scan(path):
    bad, n_kern, n_shift = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in device_objects(path, tmp):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            name, body = None, []
            funcs = []
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if name is not None: funcs.append((name, body))
                    name, body = m.group(1), []
                elif name is not None:
                    body.append(line.split("//")[0])
            if name is not None: funcs.append((name, body))
            for name, body in funcs:
                used = set()
                for ln in body:
                    for a, b, c in REG.findall(ln):
                        if a: used.add(int(a))
                        else: used.update(range(int(b), int(c) + 1))
                n_kern += 1
                for ln in body:
                    m = SHIFT.search(ln)
                    if not m: continue
                    n_shift += 1
                    k = int(m.group(2))
                    if k % 8 == 7 and (k + 1) not in used:
                        bad.append((os.path.basename(path), name, ln.strip()))
    return bad, n_kern, n_shift


def main(paths):
    all_bad, empty = [], []
    for p in paths:
        bad, nk, ns = scan(p)
        print(f"{p}: {nk} functions, {ns} 64-bit shifts by a register, {len(bad)} with the amount in the last allocated VGPR")
        all_bad += bad
        if nk == 0:   # nothing was disassembled (no device code extracted): the guard must not pass on an empty scan
            empty.append(p)
    for f, k, ln in all_bad:
        print(f"  BAD {f}: {k}: {ln}")
    for p in empty:
        print(f"  EMPTY {p}: no device function found -- not a gfx950 object / library, or llvm-objdump --offloading extracted nothing")
    return 1 if (all_bad or empty) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "obia_amd", "csrc", "libobia_hip.so")]))
