#!/usr/bin/env python3
"""Static check of the gfx950 MFMA register hazards described in vllm_rs_amd/csrc/common.cuh.

For every v_mfma in the given .s files:
  (1) its destination must not overlap SrcA or SrcB;
  (2) within the next 12 wait states nothing but an MFMA accumulating into the SAME destination may
      read or write the destination registers (the inline-asm form is invisible to hipcc's hazard
      recogniser, so the code places VRA_MFMA_DRAIN() = `s_nop 7; s_nop 4` itself).
Exit status 1 if any violation is found (the Makefile fails the build).

Also a performance lint (round 4): (3) no kernel may fetch KERNEL ARGUMENTS with vector loads.  A struct field selected under a
per-lane index (`a.seg[i].out` with i varying across lanes) makes hipcc select the field's ADDRESS inside the kernel-argument block
(s[0:1]) and read it with global_load — followed by `s_waitcnt vmcnt(0)`, which also waits for every weight tile in flight: a full HBM
round trip per occurrence.  Wave-uniform indices (readfirstlane) give scalar selects and s_loads.
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


KERNARG_VEC = re.compile(r"v_lshl_add_u64\s+v\[\d+:\d+\],\s*s\[0:1\]|global_load_\w+\s+\S+,\s*v\d+,\s*s\[0:1\]")


# tolerated: the peer-pointer table of the one-shot all-reduce is indexed by the thread on purpose (one load per launch); the
# prologue of kernel B and the epilogue wave of kernel A (int4) — both off the decode path since kernels E / W — pay one round trip
KERNARG_VEC_OK = ("oneshot_all_reduce_kernel", "gemm_skinny_kernel", "gemv_q4_kernel")
SDST = re.compile(r"^s_\w+\s+(s\[(\d+):(\d+)\]|s(\d+))(?=[,\s]|$)")


def check_kernarg_vector_loads(path):
    """s[0:1] holds the kernel-argument pointer from entry until something else is written to it"""
    bad = 0
    for fn, body in instrs(path):
        n = 0
        for t in body:
            if KERNARG_VEC.search(t):
                n += 1
            m = SDST.match(t)
            if m and not t.startswith(("s_cmp", "s_cbranch", "s_waitcnt", "s_nop", "s_barrier", "s_bitcmp")):
                lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) is not None else (int(m.group(4)), int(m.group(4)))
                if lo <= 1 and hi >= 0:
                    break  # s0 / s1 rewritten: no longer the kernel-argument pointer
        if n and any(k in fn for k in KERNARG_VEC_OK):
            print(f"[note: kernel arguments fetched with {n} vector loads, tolerated] {fn}", file=sys.stderr)
        elif n:
            bad += n
            print(f"[kernel arguments fetched with {n} vector loads] {fn}", file=sys.stderr)
    return bad


def main(paths):
    T = B = K = 0
    for p in paths:
        t, b = check(p)
        T += t
        B += b
        K += check_kernarg_vector_loads(p)
    print(f"check_mfma_overlap: {T} MFMA instructions, {B} violations" + (f", {K} vector loads of kernel arguments" if K else ""))
    return 1 if B or K else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
