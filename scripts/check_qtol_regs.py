"""Build-time check for lqr_qtol_impl.h: the accumulation registers a184..a255 hold the level-ahead pieces and are managed by inline assembly
only — no compiler-generated instruction of the code object may name them.  Usage: check_qtol_regs.py file.s [first=184]"""
import re
import sys

first = int(sys.argv[2]) if len(sys.argv) > 2 else 184
inside = False
bad = []
uses_inside = 0
for ln, line in enumerate(open(sys.argv[1]), 1):
    t = line.strip()
    if t.startswith(";;#ASMSTART"):
        inside = True
        continue
    if t.startswith(";;#ASMEND"):
        inside = False
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    code = t.split(";")[0]
    regs = []
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", code):
        regs.append((int(m.group(1)), int(m.group(2))))
    for m in re.finditer(r"\ba(\d+)\b", code):
        regs.append((int(m.group(1)), int(m.group(1))))
    hit = any(hi >= first for lo, hi in regs)
    if hit and inside:
        uses_inside += 1
    elif hit:
        bad.append((ln, t))
if bad:
    print(f"{sys.argv[1]}: {len(bad)} compiler-generated instruction(s) touch a{first}..a255:")
    for ln, t in bad[:20]:
        print(f"  line {ln}: {t}")
    sys.exit(1)
if uses_inside == 0:
    print(f"{sys.argv[1]}: no inline-assembly use of a{first}..a255 found — wrong file?")
    sys.exit(1)
print(f"{sys.argv[1]}: ok ({uses_inside} inline-assembly uses of a{first}..a255, none outside)")
