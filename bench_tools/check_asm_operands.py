#!/usr/bin/env python3
"""Guard for the generated inline-asm multipliers (field_mul_asm.cuh, field29_mul_asm.cuh): they use FIXED scratch registers
declared as clobbers.  In a compiled .s (hipcc -save-temps) every asm block is matched against its template and the registers
the compiler chose for the operands are extracted; an operand placed in a clobbered scratch register would be overwritten
mid-block.  Usage: check_asm_operands.py <file.s>; exit status 1 on an overlap."""
import re
import sys
import os
_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lurk_beta_amd', 'csrc')
hdrs = [os.path.join(_CSRC, 'field29_mul_asm.cuh'), os.path.join(_CSRC, 'field_mul_asm.cuh')]
templates = []
for h in hdrs:
    src = open(h).read()
    for m in re.finditer(r'asm(?:\s+volatile)?\(\s*((?:\s*"[^"\n]*"\s*\n?)+)\s*:\s*([^;]*?);', src, re.S):
        lines = [l for l in re.findall(r'"([^"\n]*)"', m.group(1))]
        lines = [l.replace('\\n\\t', '').replace('\\n', '').strip() for l in lines]
        lines = [l for l in lines if l]
        cl = re.findall(r'"([vs]\d+)"', m.group(2).split(':')[-1])
        templates.append((lines, set(cl)))
print("templates:", [(len(t[0]), len(t[1])) for t in templates])
s = open(sys.argv[1]).read()
blocks = re.findall(r';;#ASMSTART\n(.*?);;#ASMEND', s, re.S)
print("asm blocks in .s:", len(blocks))
bad = 0
matched = 0
for b in blocks:
    bl = [l.strip() for l in b.strip().split('\n') if l.strip()]
    cand = [t for t in templates if len(t[0]) == len(bl)]
    if not cand:
        continue
    tl, clob = cand[0]
    # try each candidate with matching first-line mnemonics
    for tl, clob in cand:
        ok = True
        ops = {}
        for a, e in zip(tl, bl):
            pat = re.escape(a)
            pat = re.sub(r'%(\d+)', lambda m: r'(?P<o' + m.group(1) + r'>\S+?)', pat.replace('\\%', '%'))
            # the same placeholder may occur twice on a line: make later ones backrefs
            seen = set()
            def fix(m):
                n = m.group(1)
                if n in seen: return '(?P=o%s)' % n
                seen.add(n); return m.group(0)
            pat = re.sub(r'\(\?P<o(\d+)>\\S\+\?\)', fix, pat)
            mm = re.fullmatch(pat, e)
            if not mm: ok = False; break
            ops.update(mm.groupdict())
        if ok:
            break
    if not ok:
        continue
    matched += 1
    clobnums = {(c[0], int(c[1:])) for c in clob}
    for k, r in ops.items():
        r = r.rstrip(',')
        m1 = re.fullmatch(r'([vs])(\d+)', r); m2 = re.fullmatch(r'([vs])\[(\d+):(\d+)\]', r)
        regs = []
        if m1: regs = [(m1.group(1), int(m1.group(2)))]
        elif m2: regs = [(m2.group(1), i) for i in range(int(m2.group(2)), int(m2.group(3)) + 1)]
        if any(x in clobnums for x in regs):
            bad += 1
            print("OVERLAP: operand %s -> %s is a clobbered scratch register" % (k, r))
print("matched blocks:", matched, "overlaps:", bad)
sys.exit(1 if bad or not matched else 0)
