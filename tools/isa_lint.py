#!/usr/bin/env python3
"""ISA lint of the shipped gfx950 code objects: what the split-bf16 kernels must not contain, as a CHECKED invariant.

Background (profiles/r06_mfma_hazard_probe.txt).  Round 5 saw run-to-run different tiles from the edge kernels once their
32 x 32 products ran as split-bf16 matrix instructions, blamed a load landing in a source register of `v_mfma_f32_16x16x32_bf16`, and fenced the
products.  Round 6 measured the hardware (tools/dev/mfma_war_probe.hip) and bisected the failing build:
  * a source register of the matrix instruction may be overwritten in the very next issue slot by a VALU instruction, an LDS read or a
    global load -- 0 wrong results in 10^9 probes, with and without a second wave on the SIMD; fenced or unfenced made no difference once
  * the packed fp32 instructions were gone: WITH `v_pk_add / mul / fma_f32` in the kernel, schedules that mix loads or matrix instructions
    into the products computed the lane-per-edge geometry wrong in lanes 48..63 of ~0.5 % of the 64-edge chunks (a `v_pk_add_f32` with
    op_sel returned src0 - 0 from sources that read back correct; the scalar instruction in its place is exact); WITHOUT them every
    schedule tried is exact and bit-identical over 5 runs x 16 configurations.
So the rule this tool FAILS on is:

  RULE 1  no kernel of the library contains packed fp32 arithmetic (`v_pk_add_f32`, `v_pk_mul_f32`, `v_pk_fma_f32`).  The defect was seen
          in the two kernels that hold multi-register-operand matrix instructions (the split-bf16 products); its mechanism is not
          understood, so the rule covers all of them.  csrc/Makefile builds with `-target-feature -packed-fp32-ops` for that.

It also REPORTS (does not fail on; `--strict-overwrite` turns it back into a failure) the older statistic: loads whose destination
overlaps an A, B or C source of such an instruction within `--min-cycles` issue cycles behind it (an s_nop N counts N + 1, a matrix
instruction 4, everything else 1), following fall-through and branch targets.  The forward kernel keeps its products as closed groups
(0 such pairs); the backward does not (the scheduler overlaps the products with the surrounding loads: ~200 pairs, exact results).

Usage:  python tools/isa_lint.py [--min-cycles 8] [--strict-overwrite] [--json out.json] [objects or shared libraries ...]
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
_PACKED_F32 = re.compile(r"^v_pk_(add|mul|fma)_f32")


def packed_fp32(name, ins):
    """RULE 1: the packed fp32 arithmetic instructions of a kernel (`beside_split_products`: the kernel also holds multi-register-operand
    matrix instructions -- the combination the defect was seen in)."""
    multi = any(m.startswith(("v_mfma", "v_smfmac")) and len(o) >= 4 and (len(regs(o[1])) > 1 or len(regs(o[2])) > 1) for _, m, o, _ in ins)
    return [{"kernel": name, "rule": "packed_fp32", "beside_split_products": multi, "instruction": f"{m} {', '.join(o)}", "addr": hex(a or 0)}
            for a, m, o, _ in ins if _PACKED_F32.match(m)]


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
    out = {"file": os.path.relpath(path, ROOT), "kernels": {}, "violations": [], "packed_fp32": []}
    for name, ins in parse(text).items():
        out["packed_fp32"] += packed_fp32(name, ins)
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
    ap.add_argument("--strict-overwrite", action="store_true", help="also fail on loads into matrix-instruction sources (the round-5 rule)")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    reports = run(a.paths, a.min_cycles)
    n_p = sum(len(r["packed_fp32"]) for r in reports)
    n_v = sum(len(r["violations"]) for r in reports)
    for r in reports:
        km = {k: v for k, v in r["kernels"].items() if v["mfma_multi_operand"]}
        print(f"{r['file']}: {len(r['kernels'])} kernels with MFMAs, {len(km)} with multi-register-operand MFMAs "
              f"({sum(v['mfma_multi_operand'] for v in km.values())} instructions); packed fp32 instructions: {len(r['packed_fp32'])} ({sum(v['beside_split_products'] for v in r['packed_fp32'])} beside split products); "
              f"loads into their sources within {a.min_cycles} cycles (reported): {len(r['violations'])}")
        for v in r["packed_fp32"][:10]:
            print(f"    PACKED FP32 {v['kernel'][:60]}: {v['instruction']} @ {v['addr']}")
        if a.strict_overwrite:
            for v in r["violations"][:20]:
                print(f"    OVERWRITE {v['kernel'][:60]}: {v['mfma']} @ {v['mfma_addr']}  <-  {v['load']} @ {v['load_addr']} after {v['issue_cycles_between']} cycles")
    if a.json:
        json.dump({"min_cycles": a.min_cycles, "reports": reports}, open(a.json, "w"), indent=1)
    print(f"isa_lint: {n_p} packed fp32 instruction(s) (RULE 1), {n_v} source-overwriting load(s) "
          f"within {a.min_cycles} issue cycles ({'checked' if a.strict_overwrite else 'reported only'}), {len(reports)} code object(s)")
    return 1 if (n_p or (a.strict_overwrite and n_v)) else 0


if __name__ == "__main__":
    sys.exit(main())
