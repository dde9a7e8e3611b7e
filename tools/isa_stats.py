"""Instruction-class counts per kernel of a hipcc -save-temps assembly file: python tools/isa_stats.py file.s substr [substr ...]
(static counts over the whole kernel body, all paths; a quick check of what the compiler made of a hot loop)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_Z\S+:)', s)
for sub in sys.argv[2:]:
    for f in funcs:
        head = f.split('\n', 1)[0]
        if sub in head:
            ins = [l.strip().split()[0] for l in f.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
            c = Counter(ins)
            grp = lambda pre, excl=(): sum(n for k, n in c.items() if k.startswith(pre) and not k.startswith(excl))
            print(head[:100])
            print(f"  total {len(ins)}  valu {grp('v_', ('v_mfma',))}  mfma {grp('v_mfma')}  ds {grp('ds_')}  s_load {grp('s_load')}  global {grp('global_')}  scratch {grp('scratch_')}  barrier {grp('s_barrier')}")
            for key in ('NumVgprs', 'NumAgprs', 'NumSgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize'):
                m = re.search(r'; ' + key + r': (\d+)', f)
                if m:
                    print(f"  {key} {m.group(1)}", end='')
            print()
            print("  ", {k: v for k, v in sorted(c.items()) if k.startswith(('v_max', 'v_exp', 'v_cndmask', 'v_cvt', 'v_add_f32', 'v_sub_f32', 'v_mul_f32', 'v_fma', 'v_pk', 'v_bfe', 'v_and', 'v_bfi', 'v_or', 'v_perm', 'v_lshl', 'v_mov'))})
