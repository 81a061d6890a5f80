#!/usr/bin/env python3
"""instruction-class sequence between workgroup barriers of one kernel instantiation in a hipcc -save-temps .s file
   (developer tool).  M mfma, T transcendental, v other VALU, L LDS, G global, w waitcnt, | barrier, n nop, s scalar."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
syms = re.findall(r'^(_Z\S+):', txt, re.M)
dem = subprocess.run(['c++filt'], input='\n'.join(syms), capture_output=True, text=True).stdout.split('\n')
target = [s for s, d in zip(syms, dem) if pat in d][0]
i = txt.index('\n' + target + ':'); j = txt.index('s_endpgm', i)
ops = [l.split()[0] for l in txt[i:j].split('\n') if l.startswith('\t') and l.split() and not l.split()[0].startswith(('.', ';'))]
def cls(o):
    if 'mfma' in o: return 'M'
    if o.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt')): return 'T'
    if o.startswith('v_'): return 'v'
    if o.startswith('ds_'): return 'L'
    if o.startswith(('global_', 'buffer_', 'scratch_')): return 'G'
    if o.startswith('s_waitcnt'): return 'w'
    if o.startswith('s_barrier'): return '|'
    if o.startswith('s_nop'): return 'n'
    return 's'
seq = ''.join(cls(o) for o in ops)
print(len(ops), "instructions")
for k, seg in enumerate(seq.split('|')):
    print(k, len(seg), seg if len(sys.argv) > 3 else seg[:400])
