#!/usr/bin/env python3
"""Instruction mix of the main loop of a kernel in a device assembly listing (hipcc -S --cuda-device-only).
Usage: isa_mix.py file.s substring-of-mangled-name ...   (prints whole-kernel and hottest-loop counts)"""
import re
import sys
from collections import Counter


def classify(i):
    if i.startswith('v_') and 'f64' in i:
        return 'valu_f64'
    if i.startswith('v_pk'):
        return 'valu_pk'
    if i.startswith('v_'):
        return 'valu'
    if i.startswith('ds_'):
        return 'lds'
    if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if i.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    lines = open(sys.argv[1]).read().split('\n')
    for sub in sys.argv[2:]:
        start = next(k for k, l in enumerate(lines) if l.startswith('_Z') and sub in l and l.rstrip().endswith(tuple([':'])) or (l.startswith('_Z') and sub in l and ': ' in l))
        end = next(k for k in range(start, len(lines)) if 's_endpgm' in lines[k])
        body = lines[start + 1:end]
        labels, ins = {}, []
        for l in body:
            t = l.strip()
            if not t or t.startswith((';', '.s', '.p', '.a', '.t', '.g', '.w', '.c')):
                continue
            if t.endswith(':') or re.match(r'^\.LBB\d+_\d+:', t):
                labels[t.split(':')[0]] = len(ins)
                continue
            ins.append(t)
        total = Counter(classify(i.split()[0]) for i in ins)
        # loops = backward branches; report the largest-trip candidate = the longest backward span
        loops = []
        for k, i in enumerate(ins):
            m = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', i)
            if m:
                tgt = m.group(1) or m.group(2)
                if tgt in labels and labels[tgt] <= k:
                    loops.append((k - labels[tgt], labels[tgt], k))
        print(sub, 'total', len(ins), dict(total))
        for span, a, b in sorted(loops, reverse=True)[:8]:
            c = Counter(classify(i.split()[0]) for i in ins[a:b + 1])
            print('   loop span', span, dict(c))


if __name__ == '__main__':
    main()
