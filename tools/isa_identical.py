#!/usr/bin/env python3
"""Which kernels of the library changed between a commit and the working tree (CPU only).

Compiles `optiland_amd/csrc/<tu>` of the given commit (a throw-away git worktree) with the
product's flags, extracts the gfx950 code object of that build and of the current in-tree object
(`optiland_amd/lib/<tu>.o`, i.e. build first), disassembles both and compares every kernel
instruction by instruction.  Answers "what does an opt-in code path cost the kernels that never
take it" without a GPU: identical ISA costs nothing.

usage: isa_identical.py [COMMIT=HEAD] [TU=trace_kernel_f32.hip]
"""

from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
         "-fno-math-errno"]
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def device_asm(obj: str, work: str) -> dict:
    """mangled kernel name -> instruction texts (addresses and encodings dropped)."""
    local = os.path.join(work, os.path.basename(obj))
    if os.path.abspath(obj) != os.path.abspath(local):
        shutil.copy(obj, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True,
                   capture_output=True, cwd=work)
    co = f"{local}.0.{TARGET}"
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True,
                          text=True).stdout
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_Z\w+)>:", ln)
        if m:
            cur = out[m.group(1)] = []
        elif cur is not None and ln.strip() and not ln.startswith("Disassembly"):
            cur.append(ln.split("//")[0].strip())
    return out


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, out.stdout.splitlines()))


def main():
    commit = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
    tu = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel_f32.hip"
    new_obj = os.path.join(ROOT, "optiland_amd", "lib", tu.replace(".hip", ".o"))
    if not os.path.exists(new_obj):
        sys.exit(f"{new_obj} missing: build the library first")
    with tempfile.TemporaryDirectory() as tmp:
        tree = os.path.join(tmp, "tree")
        subprocess.run(["git", "-C", ROOT, "worktree", "add", "--detach", "-f", tree, commit],
                       check=True, capture_output=True)
        try:
            old_obj = os.path.join(tmp, "old.o")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c",
                            os.path.join(tree, "optiland_amd", "csrc", tu), "-o", old_obj],
                           check=True, capture_output=True)
        finally:
            subprocess.run(["git", "-C", ROOT, "worktree", "remove", "--force", tree],
                           capture_output=True)
        os.makedirs(os.path.join(tmp, "a"))
        os.makedirs(os.path.join(tmp, "b"))
        old = device_asm(old_obj, os.path.join(tmp, "a"))
        new = device_asm(new_obj, os.path.join(tmp, "b"))
    rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", commit], capture_output=True,
                         text=True).stdout.strip()
    names = demangle(sorted(set(old) | set(new)))
    same = [n for n in old if n in new and old[n] == new[n]]
    changed = [n for n in old if n in new and old[n] != new[n]]
    print(f"# {tu}: gfx950 ISA of commit {rev} against the working tree's object")
    print(f"identical: {len(same)}   changed: {len(changed)}   "
          f"removed: {len([n for n in old if n not in new])}   "
          f"new: {len([n for n in new if n not in old])}")
    for n in changed:
        # same multiset of opcodes = the scheduler ordered independent instructions differently
        # (register names follow): not a change of the work done
        # (_e32 / _e64: the same opcode in its short / long encoding)
        ops_old = sorted(re.sub(r"_e(32|64)$", "", x.split()[0]) for x in old[n])
        ops_new = sorted(re.sub(r"_e(32|64)$", "", x.split()[0]) for x in new[n])
        lines = sum(1 for x, y in zip(old[n], new[n]) if x != y)
        kind = f"reordered ({lines} lines differ, same opcodes)" if ops_old == ops_new \
            else "changed"
        print(f"  {kind}  {len(old[n]):6d} -> {len(new[n]):6d} instructions  "
              f"{names[n].split('(')[0]}")
    for n in new:
        if n not in old:
            print(f"  new      {len(new[n]):6d} instructions  {names[n].split('(')[0]}")
    for n in old:
        if n not in new:
            print(f"  removed  {len(old[n]):6d} instructions  {names[n].split('(')[0]}")


if __name__ == "__main__":
    main()
