#!/usr/bin/env python3
"""Hazard lint for the hand-written (inline-asm) VMEM stores of the gfx950 kernels, on the EMITTED ISA.

LLVM's hazard recognizer inserts the wait states gfx9/CDNA needs between dependent instructions -- but it does not look
at the USES inside an inline-asm statement, and it cannot see what the asm's last instruction needs from the code after
it.  Two hazards of the `global_store_dwordx4 ... sc1` stores in persist.hip / exact.hip are therefore the source's job:

  H1  VALU writes an SGPR (v_readlane_b32 / v_readfirstlane_b32 -- how a spilled SGPR comes back --, v_cmp, carry-outs)
      -> a VMEM instruction reads that SGPR as its address: 5 wait states.  Round 3's diagnostic build of k_cg_persist had
      `v_readlane_b32 s88` two instructions before `global_store_dwordx4 v4, v[12:15], s[88:89] sc1`: a stale base, a wild
      address, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.
  H2  a VMEM store of more than 8 bytes -> the next instruction overwrites the store's DATA registers: 2 wait states on
      gfx940+.  Commit 78fa2a0: K came out wrong at 1M triangles because the instruction after the store rewrote v[..].

The lint reads `hipcc -S --offload-device-only` output and reports, per file:
  * structural: every ;;#ASMSTART block that holds a VMEM store with an SGPR base starts with >= 5 wait states of s_nop;
    every block that holds a dwordx3/x4 store ends with >= 2 wait states of s_nop;
  * contextual (what the structural rule protects against, checked on the code the compiler actually put around the
    asm): H1 and H2 instances within the hazard window, counting s_nop N as N + 1 wait states, every other
    instruction as 1, stopping at labels (a label inside the window is reported as a violation only if the window is
    not already covered by the asm's own s_nops).

Usage: isa_lint.py file.s [file.s ...]      exit code 1 when anything is reported.
"""
import re
import sys

SGPR = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b|\b(vcc)\b")
VGPR = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
STORE = re.compile(r"^(global|buffer|flat|scratch)_store_(dword(?:x[234])?|b(?:32|64|96|128)|byte|short)")


def sregs(tok):
    out = set()
    for m in SGPR.finditer(tok):
        if m.group(4):
            out |= {"vcc"}
        elif m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def vregs(tok):
    out = set()
    for m in VGPR.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse(path):
    """-> list of (kind, opcode, operands, raw, in_asm, lineno); kind in {'ins', 'label', 'asmstart', 'asmend'}"""
    out, in_asm = [], False
    for no, raw in enumerate(open(path), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            out.append(("asmstart", "", [], line, True, no))
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            out.append(("asmend", "", [], line, False, no))
            continue
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") and not line.endswith(":"):
            continue
        if line.endswith(":"):
            out.append(("label", line[:-1], [], line, in_asm, no))
            continue
        parts = line.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        out.append(("ins", parts[0], ops, line, in_asm, no))
    return out


def wait_states(op, ops):
    if op == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def valu_sgpr_defs(op, ops):
    """SGPRs a VALU instruction writes"""
    if not op.startswith("v_"):
        return set()
    if op.startswith(("v_readlane_b32", "v_readfirstlane_b32")):
        return sregs(ops[0])
    if op.startswith(("v_cmp", "v_cmpx")):
        d = sregs(ops[0]) if ops and ops[0].startswith(("s", "vcc")) else {"vcc"}
        return d
    if "_co_" in op or op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32", "v_addc", "v_subb")):
        return sregs(ops[1]) if len(ops) > 1 and ops[1].startswith(("s", "vcc")) else set()
    return set()


def vgpr_defs(op, ops):
    """VGPRs an instruction overwrites (first operand of VALU / loads); stores, branches, SALU write none"""
    if STORE.match(op) or op.startswith(("s_", "ds_write", "ds_store", "buffer_store")):
        return set()
    if op.startswith(("v_", "global_load", "buffer_load", "flat_load", "scratch_load", "ds_read", "ds_load", "ds_bpermute",
                      "ds_permute", "ds_swizzle")):
        d = vregs(ops[0]) if ops and ops[0].startswith("v") else set()
        if op.startswith(("v_swap", "v_permlane")) and len(ops) > 1:
            d |= vregs(ops[1])
        return d
    return set()


def lint(path):
    ins = parse(path)
    problems = []
    n = len(ins)
    i = 0
    while i < n:
        kind = ins[i][0]
        if kind != "asmstart":
            i += 1
            continue
        j = i + 1
        while j < n and ins[j][0] != "asmend":
            j += 1
        block = [x for x in ins[i + 1:j] if x[0] == "ins"]
        stores = [(k, x) for k, x in enumerate(block) if STORE.match(x[1])]
        if stores:
            # ---- structural
            first_k, first = stores[0]
            lead = sum(wait_states(x[1], x[2]) for x in block[:first_k] if x[1] == "s_nop")
            has_sbase = any(len(x[2]) >= 3 and x[2][2].split()[0].startswith("s[") for _, x in stores)
            if has_sbase and lead < 5:
                problems.append(f"{path}:{first[5]}: H1 structural: asm store with an SGPR base has {lead} leading wait states (< 5): {first[3]}")
            last_k, last = stores[-1]
            wide = any(re.search(r"dwordx[34]|b96|b128", x[1]) for _, x in stores)
            trail = sum(wait_states(x[1], x[2]) for x in block[last_k + 1:] if x[1] == "s_nop")
            if wide and trail < 2:
                problems.append(f"{path}:{last[5]}: H2 structural: wide asm store is followed by {trail} wait states inside its asm (< 2): {last[3]}")
            # ---- contextual H1: walk back from the first store over the instruction stream
            for _, st in stores[:1]:
                base = sregs(st[2][2].split()[0]) if len(st[2]) >= 3 and st[2][2].split()[0].startswith(("s[", "vcc")) else set()
                if base:
                    ws, k = 0, ins.index(st) - 1
                    while k >= 0 and ws < 5:
                        kd, op, ops = ins[k][0], ins[k][1], ins[k][2]
                        if kd == "label":
                            problems.append(f"{path}:{st[5]}: H1: a label {ws} wait states before the store: predecessors unknown: {st[3]}")
                            break
                        if kd == "ins":
                            hit = valu_sgpr_defs(op, ops) & base
                            if hit:
                                problems.append(f"{path}:{st[5]}: H1: `{ins[k][3]}` {ws} wait states before `{st[3]}`")
                                break
                            ws += wait_states(op, ops)
                        k -= 1
            # ---- contextual H2: walk forward from the last wide store
            if wide:
                data = vregs(last[2][1])
                ws, k = 0, ins.index(last) + 1
                while k < n and ws < 2:
                    kd, op, ops = ins[k][0], ins[k][1], ins[k][2]
                    if kd == "ins":
                        if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                            if op != "s_endpgm":
                                problems.append(f"{path}:{last[5]}: H2: a branch {ws} wait states after the store: successors unknown: {last[3]}")
                            break
                        if vgpr_defs(op, ops) & data:
                            problems.append(f"{path}:{last[5]}: H2: `{ins[k][3]}` {ws} wait states after `{last[3]}`")
                            break
                        ws += wait_states(op, ops)
                    k += 1
        i = j + 1
    return problems


def count_asm_stores(path):
    ins, c, inside = parse(path), 0, False
    for x in ins:
        if x[0] == "asmstart":
            inside = True
        elif x[0] == "asmend":
            inside = False
        elif inside and x[0] == "ins" and STORE.match(x[1]):
            c += 1
    return c


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        pr = lint(p)
        print(f"{p}: {count_asm_stores(p)} inline-asm stores, {len(pr)} problems")
        bad += pr
    for b in bad:
        print(b)
    sys.exit(1 if bad else 0)
