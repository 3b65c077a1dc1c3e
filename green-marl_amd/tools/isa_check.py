#!/usr/bin/env python3
"""isa_check.py -- verifies the hand-counted `s_waitcnt vmcnt(N)` waits of pr_cold_tile_kernel on the emitted ISA.

gmx_pr_cold.hip prefetches the entries of the next blocks with inline-asm loads into named register sets and waits for
them with inline-asm `s_waitcnt vmcnt(N)`, N counted by hand from "VMEM operations retire in issue order and every
block issues exactly 8 stores".  That is only right while the compiler emits what the source says.  This tool
compiles the file to gfx950 assembly (no GPU needed) and proves, per kernel instantiation, on the control-flow graph:

  * no scratch (a spilled prefetch register would be read before its load lands) and at most 128 VGPRs;
  * at every hand-written wait W(N) and for every prefetch set X whose registers are touched afterwards: on EVERY
    path from the asm load of X to W at least N VMEM operations were issued after the load -- so "at most N
    outstanding" implies X has landed (a dropped or merged store would make the count too small);
  * no instruction outside the asm statements mentions a register of a set between its load and the wait that
    covers it (no copy, spill or reuse of a register whose load is in flight).

Usage: isa_check.py [file.hip]   (exit code 0 = all kernels pass; prints one line per kernel)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CAP = 64          # vmcnt is a 6-bit counter: counts saturate here
INF = 10 ** 6

VMEM = re.compile(r"^(global|flat|buffer|scratch)_(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def assemble(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.dirname(src), "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def kernel_bodies(lines, prefix):
    """{mangled name: (body lines, metadata dict)} of the kernels whose mangled name starts with prefix."""
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(%s\w*):" % re.escape(prefix), l)
        if not m:
            continue
        name = m.group(1)
        end = next(j for j in range(i, len(lines)) if lines[j].strip() == "s_endpgm")
        # the last s_endpgm of the function: keep scanning until the .amdhsa_kernel directive block
        stop = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(".amdhsa_kernel"))
        body = lines[i + 1:stop]
        meta = {}
        for j in range(stop, len(lines)):
            t = lines[j].strip()
            if t.startswith(".end_amdhsa_kernel"):
                break
            mm = re.match(r"\.amdhsa_(\w+)\s+(\S+)", t)
            if mm:
                meta[mm.group(1)] = mm.group(2)
        # the resolved register / scratch numbers are printed as comments behind the kernel
        for j in range(stop, min(len(lines), stop + 400)):
            mm = re.match(r";\s*(NumVgprs|ScratchSize|NumAgprs|TotalNumVgprs|Occupancy|LDSByteSize):\s*(\d+)", lines[j].strip())
            if mm:
                meta.setdefault(mm.group(1), int(mm.group(2)))
            if "Occupancy" in meta and "LDSByteSize" in meta:
                break
        out[name] = (body, meta)
        del end
    return out


class Ins:
    __slots__ = ("kind", "text", "n", "regs", "target")

    def __init__(self, kind, text, n=0, regs=(), target=None):
        self.kind, self.text, self.n, self.regs, self.target = kind, text, n, frozenset(regs), target


def parse(body):
    """-> list of basic blocks: each {label, ins: [Ins], succ: [block index]}"""
    blocks = [{"label": None, "ins": [], "succ": []}]
    asm = None
    for raw in body:
        t = raw.split(";;#")[0].strip() if ";;#" not in raw else raw.strip()
        if t.startswith(";;#ASMSTART"):
            asm = []
            continue
        if t.startswith(";;#ASMEND"):
            joined = " ".join(asm)
            if "global_load" in joined:
                dest = set()
                nload = 0
                for a in asm:
                    for piece in a.split("\n"):
                        piece = piece.strip()
                        if piece.startswith("global_load"):
                            nload += 1
                            dest |= regs_of(piece.split(",")[0])
                blocks[-1]["ins"].append(Ins("L", joined, nload, dest))
            elif "s_waitcnt" in joined:
                blocks[-1]["ins"].append(Ins("W", joined, int(re.search(r"vmcnt\((\d+)\)", joined).group(1))))
            else:
                raise SystemExit("unknown inline asm: " + joined)
            asm = None
            continue
        if asm is not None:
            if t and not t.startswith(";"):
                asm.append(t)
            continue
        t = t.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                blocks.append({"label": m.group(1), "ins": [], "succ": []})
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            blocks.append({"label": m.group(1), "ins": [], "succ": []})
            continue
        op = t.split()[0]
        if op == "s_branch" or op.startswith("s_cbranch"):
            blocks[-1]["ins"].append(Ins("B" if op == "s_branch" else "C", t, target=t.split()[-1]))
            blocks.append({"label": None, "ins": [], "succ": []})
            continue
        if op == "s_endpgm":
            blocks[-1]["ins"].append(Ins("E", t))
            blocks.append({"label": None, "ins": [], "succ": []})
            continue
        blocks[-1]["ins"].append(Ins("V" if VMEM.match(op) else "O", t, regs=regs_of(t)))
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    for i, b in enumerate(blocks):
        last = b["ins"][-1] if b["ins"] else None
        if last is not None and last.kind in ("B", "C"):
            b["succ"].append(index[last.target])
        if last is None or last.kind not in ("B", "E"):
            if i + 1 < len(blocks):
                b["succ"].append(i + 1)
    return blocks


def check_kernel(body):
    """-> (errors, stats)"""
    blocks = parse(body)
    # Every asm load STATEMENT is tracked on its own (the kernel has one two-set pipeline per work-item form, and the
    # register allocator reuses the same registers across forms): what a statement has "in flight" is the part of its
    # destination registers that no later asm load has overwritten.
    loads = [ins for b in blocks for ins in b["ins"] if ins.kind == "L"]
    if not loads:
        return ["no inline-asm prefetch found"], {}
    lid = {id(ins): i for i, ins in enumerate(loads)}
    nl = len(loads)
    # state per block entry: per load statement (VMEM operations issued since it, landed?, registers still its own); None = unreached
    entry = [None] * len(blocks)
    entry[0] = [(INF, True, frozenset())] * nl
    errors, waits = [], {}
    work = [0]

    def flow(i, report):
        st = list(entry[i])
        for ins in blocks[i]["ins"]:
            if ins.kind == "L":
                k = lid[id(ins)]
                st = [(min(CAP, c + ins.n) if c < INF else INF, r, g - ins.regs) for (c, r, g) in st]
                st[k] = (0, False, ins.regs)
            elif ins.kind == "W":
                st = [(c, r or (c < INF and c >= ins.n), g) for (c, r, g) in st]
                if report:
                    waits[ins.n] = waits.get(ins.n, 0) + 1
            elif ins.kind in ("V", "O"):
                if report:
                    for (c, r, g) in st:
                        if not r and (ins.regs & g):
                            errors.append("`%s` touches v%s of a prefetch set whose load may still be in flight (%d VMEM operations "
                                          "since the load)" % (ins.text, sorted(ins.regs & g), c))
                if ins.kind == "V":
                    st = [(min(CAP, c + 1) if c < INF else INF, r, g) for (c, r, g) in st]
        return st

    while work:
        i = work.pop()
        out = flow(i, False)
        for j in blocks[i]["succ"]:
            if entry[j] is None:
                entry[j] = list(out)
                work.append(j)
            else:
                merged = [(min(a[0], b[0]), a[1] and b[1], a[2] | b[2]) for a, b in zip(entry[j], out)]
                if merged != entry[j]:
                    entry[j] = merged
                    work.append(j)
    for i in range(len(blocks)):
        if entry[i] is not None:
            flow(i, True)
    if any(re.match(r"scratch_", ins.text) for b in blocks for ins in b["ins"]):
        errors.append("scratch instructions present")
    stats = {"asm_loads": nl, "waits": waits,
             "stores": sum(ins.text.startswith("global_store") for b in blocks for ins in b["ins"])}
    return errors, stats


def check_file(src, prefix="_Z19pr_cold_tile_kernel"):
    lines = assemble(src)
    kernels = kernel_bodies(lines, prefix)
    if not kernels:
        return {"": (["no kernel named %s* in %s" % (prefix, src)], {})}
    res = {}
    for name, (body, meta) in kernels.items():
        errors, stats = check_kernel(body)
        scratch = meta.get("ScratchSize", int(meta.get("private_segment_fixed_size", 0)))
        vgprs = meta.get("NumVgprs", -1)
        if scratch != 0:
            errors.append("scratch size %s != 0" % scratch)
        if vgprs > 128:
            errors.append("%d VGPRs > 128 (the kernel needs 4 waves per SIMD)" % vgprs)
        stats.update({"vgprs": vgprs, "scratch": scratch})
        res[name] = (errors, stats)
    return res


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "green-marl_amd", "csrc", "gmx_pr_cold.hip")
    bad = 0
    for name, (errors, stats) in check_file(src).items():
        print("%s: %s %s" % (name[:48], "FAIL" if errors else "ok", stats))
        for e in errors[:20]:
            print("   " + e)
        bad += bool(errors)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
