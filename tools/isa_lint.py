#!/usr/bin/env python3
"""ISA lint of the shipped gfx950 code objects: the MFMA operand-overwrite rule as a CHECKED invariant.

Background (profiles/r05_bf16x3_edge_kernels.txt part F): inside the split-bf16 products of the edge kernels
(`mv16_bf`, csrc/egnn_mfma16.h) the compiler once re-used an A/B operand register of a `v_mfma_f32_16x16x32_bf16` as the destination
of a global / LDS load issued right behind it; on the MI355X that produced run-to-run different tiles.  The fix makes every product a
closed instruction group (`sched_barrier` + operands kept alive over `EQ_BF_GUARD_NOPS` x `s_nop 7`).  Nothing but the generated code
says whether that still holds after a compiler bump or an edit, so this tool reads the generated code:

  for every MFMA whose A or B operand is a multi-register tuple (the bf16 / f16 / fp8 forms; 128-bit operands are what the hazard
  was seen on), walk every path that leaves the instruction (fall-through and branch targets) and require that no VMEM / LDS / flat /
  scratch LOAD whose destination overlaps one of the instruction's A, B or C source registers issues before `--min-cycles` issue
  cycles have passed (default 8 = one `s_nop 7`; an s_nop N counts N + 1, a matrix instruction 4, everything else 1).

The single-register-operand fp32 forms (`v_mfma_f32_16x16x4_f32`: A and B are one VGPR each, read in the issue cycle) are counted
and reported -- loads DO overwrite their operands at distance 1 in the shipped kernels, with bit-exact, reproducible results in every
run since round 1 -- but not failed.

Usage:  python tools/isa_lint.py [--min-cycles 8] [--json out.json] [objects or shared libraries ...]
        (default: every csrc/*.o that carries gfx950 code).  Exit code 1 on a violation.  Needs /opt/rocm/lib/llvm/bin.
Used by tests/test_isa_lint.py (CPU suite) and `make -C csrc lint`."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_REG = re.compile(r"([va])(?:\[(\d+):(\d+)\]|(\d+))")
_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load|ds_read|ds_load|tbuffer_load|global_atomic\w*_rtn|ds_\w+_rtn|ds_bpermute|ds_permute|ds_swizzle|ds_consume|ds_append)")
_BRANCH = re.compile(r"^s_c?branch")


def regs(tok):
    """'v[86:89]' -> {('v', 86) ... ('v', 89)}; 'a3' -> {('a', 3)}; anything else -> empty."""
    m = _REG.fullmatch(tok.strip())
    if not m:
        return frozenset()
    if m.group(4) is not None:
        return frozenset({(m.group(1), int(m.group(4)))})
    return frozenset((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))


_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def disassemble(path, workdir):
    """gfx950 disassembly text of an object / shared library built by hipcc, or None without device code.  An object's .hip_fatbin
    section is one offload bundle; a linked library's is the concatenation of its objects' bundles: each is unbundled on its own."""
    base = os.path.join(workdir, os.path.basename(path))
    fat = base + ".fatbin"
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", path, os.path.join(workdir, "discard.o")],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(_MAGIC), blob)]
    if not starts:
        return None
    texts = []
    for k, a in enumerate(starts):
        piece, co = f"{base}.{k}.bundle", f"{base}.{k}.co"
        with open(piece, "wb") as fh:
            fh.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={piece}",
                            f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"llvm-objdump failed on {co}: {r.stderr[-400:]}")
        texts.append(r.stdout)
    return "\n".join(texts) if texts else None


def parse(text):
    """-> {kernel: [[addr, mnemonic, [operand tokens], branch target address or None], ...]} from llvm-objdump -d text
    (instruction lines look like `\tv_mfma... v[20:23], ...   // 000000048EDC: D3B50014 04527956 <kernel+0x1ec>`)."""
    kernels, cur, base = {}, None, 0
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", ln)
        if m:
            cur, base = kernels.setdefault(m.group(2), []), int(m.group(1), 16)
            continue
        if cur is None or not ln.startswith(("\t", " ")):
            continue
        body, _, tail = ln.partition("//")
        body = body.strip()
        if not body:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        addr = int(am.group(1), 16) if am else None
        parts = body.split(None, 1)
        mnem = parts[0]
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        target = None
        if _BRANCH.match(mnem):
            tm = re.search(r"<[^>+]+\+0x([0-9a-fA-F]+)>\s*$", tail)
            if tm:
                target = base + int(tm.group(1), 16)
            elif re.search(r"<[^>+]+>\s*$", tail):
                target = base
        cur.append([addr, mnem, ops, target])
    return kernels


def cost(mnem, ops):
    if mnem == "s_nop":
        try:
            return int(ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    if mnem.startswith(("v_mfma", "v_smfmac")):
        return 4
    return 1


def lint_kernel(name, ins, min_cycles):
    """-> (violations, n multi-operand MFMAs, n fp32 single-operand MFMAs, n of those with a distance-1 overwrite, min observed distance)"""
    by_addr = {i[0]: k for k, i in enumerate(ins) if i[0] is not None}
    viol, n_multi, n_single, n_single_d1, min_seen = [], 0, 0, 0, None
    for k, (addr, mnem, ops, _) in enumerate(ins):
        if not mnem.startswith(("v_mfma", "v_smfmac")) or len(ops) < 4:
            continue
        a, b, c = regs(ops[1]), regs(ops[2]), regs(ops[3].split()[0])
        multi = len(a) > 1 or len(b) > 1
        srcs = a | b | c
        if not srcs:
            continue
        if multi:
            n_multi += 1
        else:
            n_single += 1
        # walk every path behind the instruction until `min_cycles` issue cycles have passed
        limit = min_cycles if multi else 2
        stack, seen = [(k + 1, 0)], set()
        hit_d1 = False
        while stack:
            j, cyc = stack.pop()
            while j < len(ins) and cyc < limit:
                if (j, cyc) in seen:
                    break
                seen.add((j, cyc))
                _, m2, o2, tgt = ins[j]
                if _LOAD.match(m2) and o2:
                    dst = regs(o2[0])
                    if dst & srcs:
                        if multi:
                            viol.append({"kernel": name, "mfma": f"{mnem} {', '.join(ops)}", "mfma_addr": hex(addr or 0),
                                         "load": f"{m2} {', '.join(o2)}", "load_addr": hex(ins[j][0] or 0), "issue_cycles_between": cyc,
                                         "overlap": sorted(f"{t}{r}" for t, r in dst & srcs)})
                        else:
                            hit_d1 = True
                        if min_seen is None or cyc < min_seen:
                            min_seen = cyc
                if m2 == "s_endpgm":
                    break
                if tgt is not None:
                    t_idx = by_addr.get(tgt)
                    if t_idx is not None:
                        stack.append((t_idx, cyc + 1))
                    if m2 == "s_branch":
                        break
                cyc += cost(m2, o2)
                j += 1
        if hit_d1:
            n_single_d1 += 1
    return viol, n_multi, n_single, n_single_d1, min_seen


def group_margin(ins):
    """Smallest number of issue cycles between a multi-register-operand MFMA and the next load that overwrites one of ITS sources along
    the fall-through path (None when no such load follows within 64 instructions): the margin the guard actually leaves."""
    best = None
    for k, (addr, mnem, ops, _) in enumerate(ins):
        if not mnem.startswith("v_mfma") or len(ops) < 4:
            continue
        a, b = regs(ops[1]), regs(ops[2])
        if len(a) <= 1 and len(b) <= 1:
            continue
        srcs = a | b | regs(ops[3].split()[0])
        cyc = 0
        for j in range(k + 1, min(k + 65, len(ins))):
            _, m2, o2, _ = ins[j]
            if _LOAD.match(m2) and o2 and (regs(o2[0]) & srcs):
                best = cyc if best is None else min(best, cyc)
                break
            cyc += cost(m2, o2)
    return best


def lint_file(path, min_cycles, workdir):
    text = disassemble(path, workdir)
    if text is None:
        return None
    out = {"file": os.path.relpath(path, ROOT), "kernels": {}, "violations": []}
    for name, ins in parse(text).items():
        v, n_multi, n_single, n_d1, _ = lint_kernel(name, ins, min_cycles)
        if n_multi or n_single:
            out["kernels"][name] = {"mfma_multi_operand": n_multi, "mfma_fp32_single_operand": n_single,
                                    "fp32_with_distance_le_1_overwrite": n_d1, "instructions": len(ins),
                                    "min_cycles_to_overwriting_load": group_margin(ins) if n_multi else None}
        out["violations"] += v
    return out


def default_objects():
    """The objects of the in-tree build; without them (a tree that ships only the linked library) the library itself."""
    objs = sorted(glob.glob(os.path.join(ROOT, "se3_equi_graph_registration_amd", "csrc", "*.o")))
    so = os.path.join(ROOT, "se3_equi_graph_registration_amd", "libequigspr.so")
    return objs or ([so] if os.path.exists(so) else [])


def run(paths=None, min_cycles=8):
    reports = []
    with tempfile.TemporaryDirectory() as wd:
        for p in (paths or default_objects()):
            r = lint_file(p, min_cycles, wd)
            if r is not None:
                reports.append(r)
    return reports


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--min-cycles", type=int, default=8)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    reports = run(a.paths, a.min_cycles)
    n_v = sum(len(r["violations"]) for r in reports)
    for r in reports:
        km = {k: v for k, v in r["kernels"].items() if v["mfma_multi_operand"]}
        print(f"{r['file']}: {len(r['kernels'])} kernels with MFMAs, {len(km)} with multi-register-operand MFMAs "
              f"({sum(v['mfma_multi_operand'] for v in km.values())} instructions), violations: {len(r['violations'])}; "
              f"fp32 single-operand MFMAs: {sum(v['mfma_fp32_single_operand'] for v in r['kernels'].values())} "
              f"({sum(v['fp32_with_distance_le_1_overwrite'] for v in r['kernels'].values())} with an overwriting load within 1 cycle)")
        margins = [v["min_cycles_to_overwriting_load"] for v in km.values() if v["min_cycles_to_overwriting_load"] is not None]
        if margins:
            print(f"    smallest distance from a multi-operand MFMA to a load overwriting its sources: {min(margins)} issue cycles")
        for v in r["violations"][:20]:
            print(f"    VIOLATION {v['kernel'][:60]}: {v['mfma']} @ {v['mfma_addr']}  <-  {v['load']} @ {v['load_addr']} after {v['issue_cycles_between']} cycles")
    if a.json:
        json.dump({"min_cycles": a.min_cycles, "reports": reports}, open(a.json, "w"), indent=1)
    print(f"isa_lint: {n_v} violation(s) in {len(reports)} code object(s) (rule: no load into an A/B/C source of a multi-register-operand MFMA within "
          f"{a.min_cycles} issue cycles)")
    return 1 if n_v else 0


if __name__ == "__main__":
    sys.exit(main())
