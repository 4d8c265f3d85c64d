#!/usr/bin/env python
"""Static instruction mix per basic block of a kernel in a gfx950 assembly listing (hipcc -S): the first one, or the one whose
mangled name contains `kernel`.
usage: python tools/isa_blocks.py file.s [min_valu] [kernel]"""
import re
import sys


def main(path, min_valu=25, kernel=None):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and (kernel is None or kernel in l.split(':')[0])][0]
    end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith('.Lfunc_end')][0]
    blocks = []

    def new(name, cmt):
        b = dict(name=name, cmt=cmt or '', valu=0, salu=0, mem=0, f64=0, trans=0, dpp=0, line=0)
        blocks.append(b)
        return b
    blk = new('entry', '')
    for i, l in enumerate(lines[start + 1:end]):
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l) or re.match(r'^; %(bb\.\d+):\s*(;.*)?$', l)
        if m:
            blk = new(m.group(1), m.group(2))
            blk['line'] = start + 1 + i
            continue
        t = l.strip()
        if not t or t[0] in ';.':
            continue
        op = t.split()[0]
        if op.startswith('v_'):
            blk['valu'] += 1
        elif op.startswith('s_'):
            blk['salu'] += 1
        else:
            blk['mem'] += 1
        if '_f64' in op:
            blk['f64'] += 1
        if re.search(r'(rcp|sqrt|rsq)_f64', op):
            blk['trans'] += 1
        if 'dpp' in t:
            blk['dpp'] += 1
    print('total valu', sum(b['valu'] for b in blocks), 'blocks', len(blocks))
    for b in blocks:
        if b['valu'] >= min_valu:
            print(f"{b['line']:5d} {b['name']:10s} valu {b['valu']:4d} f64 {b['f64']:4d} trans {b['trans']:2d} dpp {b['dpp']:3d} "
                  f"salu {b['salu']:3d} mem {b['mem']:3d}  {b['cmt'][:50]}")


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25, sys.argv[3] if len(sys.argv) > 3 else None)
