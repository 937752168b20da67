#!/usr/bin/env python3
"""Static check of the gfx950 MFMA register hazards described in vllm_rs_amd/csrc/common.cuh.

For every v_mfma in the given .s files:
  (1) its destination must not overlap SrcA or SrcB;
  (2) within the next 12 wait states nothing but an MFMA accumulating into the SAME destination may
      read or write the destination registers (the inline-asm form is invisible to hipcc's hazard
      recogniser, so the code places VRA_MFMA_DRAIN() = `s_nop 7; s_nop 4` itself).
Exit status 1 if any violation is found (the Makefile fails the build).
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
WAIT = 12


def regset(text):
    s = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            s.add(int(m.group(3)))
    return s


def instrs(path):
    """yield (function, [instruction lines]) with comments/labels/directives stripped"""
    cur, body = None, []
    for line in open(path, errors="ignore"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if cur and body:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        body.append(t)
        if t.startswith("s_endpgm"):
            yield cur, body
            cur, body = None, []
    if cur and body:
        yield cur, body


def check(path):
    total = bad = 0
    for fn, body in instrs(path):
        for i, t in enumerate(body):
            m = re.match(r"v_mfma_\w+\s+(\S+),\s*(\S+),\s*(\S+),\s*(\S+)", t)
            if not m:
                continue
            total += 1
            d, a, b, c = [x.strip(",") for x in m.groups()]
            D, A, B = regset(d), regset(a), regset(b)
            if D & A or D & B:
                bad += 1
                print(f"[overlap] {fn}: {t}", file=sys.stderr)
                continue
            waited = 0
            for u in body[i + 1:]:
                if waited >= WAIT:
                    break
                mm = re.match(r"v_mfma_\w+\s+(\S+),\s*(\S+),\s*(\S+),\s*(\S+)", u)
                if mm:
                    d2, a2, b2, c2 = [x.strip(",") for x in mm.groups()]
                    if regset(d2) == D and regset(c2) == D and not (regset(a2) & D) and not (regset(b2) & D):
                        break  # accumulate chain: the next MFMA takes over the tracking
                    if (regset(d2) | regset(a2) | regset(b2) | regset(c2)) & D:
                        bad += 1
                        print(f"[early use by another mfma after {waited} states] {fn}: {t}  ->  {u}", file=sys.stderr)
                        break
                    waited += 1
                    continue
                mn = re.match(r"s_nop\s+(\d+)", u)
                if mn:
                    waited += int(mn.group(1)) + 1
                    continue
                if u.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    break  # control flow: the linear scan ends here (regions are written branch free)
                if regset(u) & D:
                    bad += 1
                    print(f"[early use after {waited} states] {fn}: {t}  ->  {u}", file=sys.stderr)
                    break
                waited += 1
    return total, bad


def main(paths):
    T = B = 0
    for p in paths:
        t, b = check(p)
        T += t
        B += b
    print(f"check_mfma_overlap: {T} MFMA instructions, {B} violations")
    return 1 if B else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
