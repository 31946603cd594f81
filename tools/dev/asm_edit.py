#!/usr/bin/env python3
"""Edit the device assembly of one kernel: insert waits / idle slots around instruction classes (ISA-level bisection).
   asm_edit.py in.s out.s <kernel name substring> <rule>[,<rule>...]
   rules: wait_before_mfma, nop_after_mfma, wait_before_store, nop_before_store, wait_after_load, wait_before_lds_read, wait_after_lds_write,
          wait_before_bf16, wait_before_f32mfma, wait_before_valu"""
import re, sys
src, dst, kern, rules = sys.argv[1], sys.argv[2], sys.argv[3], set(sys.argv[4].split(","))
WAIT = "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n"
NOP = "\ts_nop 7\n"
out, inside, n = [], False, 0
for ln in open(src):
    if not inside:
        if re.match(r"^(\S+):", ln) and kern in ln.split(":")[0] and ln.startswith("_Z"):
            inside = True
        out.append(ln); continue
    if ln.startswith(".Lfunc_end"):
        inside = False
        out.append(ln); continue
    t = ln.strip()
    m = t.split()[0] if t and not t.startswith((";", ".", "//")) and not t.endswith(":") else ""
    pre, post = "", ""
    is_mfma = m.startswith("v_mfma")
    is_store = m.startswith(("ds_write", "global_store", "buffer_store", "ds_add", "global_atomic"))
    is_load = m.startswith(("ds_read", "global_load", "buffer_load"))
    if is_mfma and "wait_before_mfma" in rules: pre += WAIT
    if is_mfma and "bf16" in m and "wait_before_bf16" in rules: pre += WAIT
    if is_mfma and "bf16" not in m and "wait_before_f32mfma" in rules: pre += WAIT
    if is_mfma and "nop_after_mfma" in rules: post += NOP
    if is_store and "wait_before_store" in rules: pre += WAIT
    if is_store and "nop_before_store" in rules: pre += NOP + NOP
    if is_load and "wait_after_load" in rules: post += WAIT
    if m.startswith("ds_read") and "wait_before_lds_read" in rules: pre += WAIT
    if m.startswith("ds_write") and "wait_after_lds_write" in rules: post += WAIT
    if m.startswith("v_") and not is_mfma and "wait_before_valu" in rules: pre += WAIT
    if m.startswith(("global_store", "buffer_store")) and "wait_after_gstore" in rules: post += WAIT
    if m.startswith(("global_load", "buffer_load")) and "wait_after_gload" in rules: post += WAIT
    if m.startswith("ds_read") and "wait_after_lds_read" in rules: post += WAIT
    if is_mfma and "nop_before_mfma" in rules: pre += "\ts_nop 3\n"
    if is_mfma and "bf16" in m and "nop_before_bf16" in rules: pre += "\ts_nop 3\n"
    if ("_dpp" in m or "permlane" in m) and "nop_before_dpp" in rules: pre += "\ts_nop 3\n"
    if ("_dpp" in m or "permlane" in m) and "nop_after_dpp" in rules: post += "\ts_nop 3\n"
    if m.startswith(("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log")) and "nop_after_trans" in rules: post += "\ts_nop 3\n"
    if m.startswith("v_perm_b32") and "nop_after_perm" in rules: post += "\ts_nop 3\n"
    if m.startswith("ds_read_b128") and "nop_before_frag" in rules: pre += NOP
    if m.startswith("v_") and not is_mfma and "nop_after_valu" in rules: post += "\ts_nop 1\n"
    if m == "s_waitcnt" and "nop_after_waitcnt" in rules: post += "\ts_nop 3\n"
    if pre or post: n += 1
    out.append(pre + ln + post)
open(dst, "w").writelines(out)
print(f"{dst}: {n} instructions of `{kern}` patched with {sorted(rules)}")
