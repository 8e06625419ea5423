#!/usr/bin/env python3
"""Instruction histogram of one kernel in a gfx950 .s file (design aid).
usage: asm_hist.py file.s <substring of mangled name> [topN]"""
import collections, re, sys
def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(('):', ':')) or (l.startswith('_Z') and key in l and ':' in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    ops = collections.Counter()
    for l in lines[start:end + 1]:
        m = re.match(r'^\s+([a-z][a-z_0-9]+)\s', l)
        if m: ops[m.group(1)] += 1
    tot = sum(ops.values())
    mulc = sum(v for k, v in ops.items() if k in ('v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32', 'v_mul_u32_u24', 'v_mul_hi_u32_u24'))
    print(lines[start][:90]); print('instructions', tot, ' 32-bit-multiplier ops', mulc)
    for k, v in ops.most_common(top): print(f'  {k:28s}{v}')
main()
