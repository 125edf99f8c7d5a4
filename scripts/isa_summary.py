#!/usr/bin/env python3
"""Resource and instruction summary of one kernel in a hipcc -S listing:  isa_summary.py file.s <mangled-name-substring> [--loop]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2]
names = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if sub in n]
for name in names:
    a = s.index(name + ':')
    b = s.index('.end_amdhsa_kernel', a)
    body = s[a:b]
    print(name, 'lines', body.count('\n'))
    for key in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_next_free_sgpr',
                '.amdhsa_private_segment_fixed_size', '.amdhsa_group_segment_fixed_size'):
        m = re.search(re.escape(key) + r'.*', body)
        print('  ', m.group(0) if m else None)
    ops = collections.Counter(re.findall(r'^\s+([a-z_0-9]+)', body, re.M))
    print('  ', {k: v for k, v in ops.items() if k.startswith(('global_load', 'global_store', 'v_mfma', 's_waitcnt', 's_barrier',
                                                              'scratch', 'v_accvgpr', 's_cbranch', 'ds_', 's_nop', 'buffer_'))})
    if '--loop' in sys.argv:
        lines = body.split('\n')
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        labels = [i for i, l in enumerate(lines) if re.match(r'^\.LBB', l)]
        for li, lab in enumerate(labels):
            end = labels[li + 1] if li + 1 < len(labels) else len(lines)
            blk = lines[lab:end]
            if any('v_mfma' in l for l in blk) and any(lines[lab].split(':')[0] in l and 's_cbranch' in l for l in blk):
                print('\n'.join(blk))
