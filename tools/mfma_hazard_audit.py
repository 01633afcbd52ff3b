#!/usr/bin/env python3
"""Static audit of MFMA -> VALU wait states in the filter kernels' gfx950 ISA.

Why: the in-block rule the compiler applies for `v_mfma_f32_32x32x16_f16` (8 passes) is 12 wait
states between the MFMA and the first VALU instruction that reads (or overwrites) its result.
When the reader sat in a LATER basic block (a per-step `if (hit)` branch between them) the
inserted s_nops only made up 6, and with no other MFMA in between the accumulator was read
stale about once in 60 launches (k = 17, last query tile of a group: wrong nearest index).
The kernels now keep MFMA and reader in one block; this audit follows every path from each MFMA
(fall-through and taken branches) and reports any MFMA result touched by a VALU/memory instruction fewer than
NEED wait states later, so a future edit that re-introduces the pattern fails on the CPU.

usage: mfma_hazard_audit.py file.s [need=12]   (file.s from `hipcc -S --cuda-device-only`)
"""
import re
import sys

NEED = 12


def need_of(mnemonic, need32=NEED):
    """Wait states between an MFMA and the first VALU / memory instruction that touches its result: 12 for the 8-pass
    v_mfma_f32_32x32x16_f16 (what hipcc inserts inside a basic block), 8 for the 4-pass 16x16x32 forms (round 4: the deep-K
    scan's 16 x 16 shape; hipcc pads those with s_nop 7 = 8 states in a block of their own)."""
    return 8 if "_16x16x" in mnemonic else need32


def _regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(text, need=NEED):
    """-> (violations, number of MFMAs audited); a violation is (function, wait_states, mfma_line,
    reader_line).  Every path from an MFMA is followed — fall-through AND taken branches — until
    `need` wait states have gone by, the result has been overwritten, or the function ends."""
    ins = []          # (line, kind, text): kind FUNC | LABEL | I
    for ln, raw in enumerate(text.split("\n"), 1):
        s = raw.strip()
        if re.match(r"^_Z\w+:", s):
            ins.append((ln, "FUNC", s.split(":")[0]))
            continue
        m = re.match(r"^(\.LBB[\w.]+):", s)
        if m:
            ins.append((ln, "LABEL", m.group(1)))
            continue
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if s:
            ins.append((ln, "I", s))
    label_at = {t: i for i, (_, kind, t) in enumerate(ins) if kind == "LABEL"}
    found, cur, n_mfma = [], None, 0
    for i, (ln, kind, s) in enumerate(ins):
        if kind == "FUNC":
            cur = s
            continue
        if kind != "I" or not s.startswith("v_mfma"):
            continue
        n_mfma += 1
        dst = _regs(s.split(None, 1)[1].split(", ")[0])
        need_here = need_of(s.split()[0], need)
        work = [(i + 1, 0)]
        seen = {}
        while work:
            j, states = work.pop()
            while j < len(ins) and states < need_here:
                if seen.get(j, need_here + 1) <= states:
                    break  # reached before with no more wait states behind it
                seen[j] = states
                ln2, k2, s2 = ins[j]
                if k2 == "FUNC":
                    break
                if k2 == "LABEL":
                    j += 1
                    continue
                if s2.startswith(("s_endpgm", "s_setpc", "s_trap")):
                    break
                if s2.startswith("s_branch"):
                    t = label_at.get(s2.split()[1])
                    if t is None:
                        break
                    states += 1
                    j = t
                    continue
                if s2.startswith("s_cbranch"):
                    t = label_at.get(s2.split()[-1])
                    if t is not None:
                        work.append((t, states + 1))
                    states += 1
                    j += 1
                    continue
                if s2.startswith("v_mfma"):
                    o2 = s2.split(None, 1)[1].split(", ")
                    if _regs(o2[0]) & dst:
                        break  # accumulate chain / overwritten by the next MFMA: the XDL rule, not this one
                    states += 1
                    j += 1
                    continue
                if s2.startswith("s_nop"):
                    states += int(s2.split()[1]) + 1
                    j += 1
                    continue
                if s2.startswith(("v_", "global_", "ds_", "buffer_", "flat_", "scratch_")):
                    touched = _regs(s2.split(None, 1)[1]) if " " in s2 else set()
                    if touched & dst:
                        if states < need_here:
                            found.append((cur, states, ln, ln2))
                        break
                states += 1
                j += 1
    return sorted(set(found)), n_mfma


def audit_operands(text, need=2):
    """The other direction (round 3): a VGPR written by a VALU instruction and read by an MFMA as its
    A, B or C operand fewer than `need` wait states later (the matrix core then reads the OLD value).  hipcc pads this for
    code it schedules itself and pads NOTHING inside an inline-asm string — one arm of the C5 experiments produced wrong
    answers that way (profiles/r02_c5_variants.txt).  Walks every function in fall-through order (a label resets nothing:
    a jump target right in front of an MFMA can only add wait states on the taken path, never remove those of the
    fall-through path that is checked here).  -> (violations, MFMAs looked at); a violation is
    (function, wait_states, writer_line, mfma_line)."""
    found, cur, n_mfma = [], None, 0
    recent = []     # (line, set of VGPRs written, wait states since)
    for ln, raw in enumerate(text.split("\n"), 1):
        s = raw.strip()
        if re.match(r"^_Z\w+:", s):
            cur, recent = s.split(":")[0], []
            continue
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        if s.startswith("s_nop"):
            k = int(s.split()[1]) + 1
            recent = [(l, r, w + k) for (l, r, w) in recent if w + k < need]
            continue
        if s.startswith("v_mfma"):
            n_mfma += 1
            ops = s.split(None, 1)[1].split(", ")
            src = set()
            for o in ops[1:4]:
                src |= _regs(o)
            for (l, r, w) in recent:
                if r & src and w < need:
                    found.append((cur, w, l, ln))
            recent = [(l, r, w + 1) for (l, r, w) in recent if w + 1 < need]
            continue
        recent = [(l, r, w + 1) for (l, r, w) in recent if w + 1 < need]
        if s.startswith("v_") and not s.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")) and " " in s:
            dst = _regs(s.split(None, 1)[1].split(", ")[0])
            if dst:
                recent.append((ln, dst, 0))
    return sorted(set(found)), n_mfma


if __name__ == "__main__":
    need = int(sys.argv[2]) if len(sys.argv) > 2 else NEED
    bad, n = audit(open(sys.argv[1]).read(), need)
    for f in bad:
        print("%s: %d wait states (MFMA line %d, reader line %d)" % f)
    print("%d MFMAs audited, %d short of %d wait states" % (n, len(bad), need))
    bad2, _ = audit_operands(open(sys.argv[1]).read())
    for f in bad2:
        print("%s: operand written %d wait states before the MFMA reads it (writer line %d, MFMA line %d)" % f)
    print("%d MFMA operands written fewer than 2 wait states before the read" % len(bad2))
    sys.exit(1 if bad or bad2 else 0)
