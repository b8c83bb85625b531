"""Instruction mix / register use of one kernel in a `hipcc -save-temps` assembly file.

    python tools/isa_mix.py <file.s> <substring of the mangled kernel name> [...more substrings]
"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    keys = sys.argv[2:]
    for m in re.finditer(r'^(_Z[^\n:]*):[^\n]*\n(.*?); Occupancy: \d+', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not all(k in name for k in keys):
            continue
        vg = re.search(r'; NumVgprs: (\d+)', body)
        sc = re.search(r'; ScratchSize: (\d+)', body)
        occ = re.search(r'; Occupancy: (\d+)', m.group(0))
        ops = {}
        for line in body.split('\n'):
            t = line.strip().split(' ')[0]
            if t.startswith(('v_', 'ds_', 's_', 'global_', 'buffer_', 'scratch_')):
                ops[t] = ops.get(t, 0) + 1
        top = sorted(ops.items(), key=lambda x: -x[1])[:24]
        print(name)
        print('  vgpr', vg.group(1) if vg else None, 'scratch', sc.group(1) if sc else None,
              'occupancy', occ.group(1) if occ else None, 'instructions', sum(ops.values()))
        print('  ' + ', '.join(f'{k}:{v}' for k, v in top))


if __name__ == '__main__':
    main()
