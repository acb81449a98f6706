"""Instruction-class sequence of a line range of an assembly listing: isa_seq.py file.s first last"""
import sys
lines = open(sys.argv[1]).read().split("\n")[int(sys.argv[2]) - 1:int(sys.argv[3])]
seq = []
for l in lines:
    l = l.strip()
    if not l or l[0] in ".;":
        continue
    op = l.split()[0]
    k = ('M' if op.startswith('v_mfma') else 'r' if op.startswith('ds_read') else 'W' if op.startswith('ds_write') else
         'G' if 'load_lds' in op else 'L' if op.startswith(('global_load', 'buffer_load')) else
         'S' if op.startswith(('buffer_store', 'global_store')) else 'w' if op.startswith('s_waitcnt') else
         'B' if op.startswith('s_barrier') else '|' if op.startswith(('s_cbranch', 's_branch')) else
         'v' if op.startswith('v_') else 's')
    seq.append(k)
print(''.join(seq))
