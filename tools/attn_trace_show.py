"""Per-decode-step means of the durations written by tools/attn_trace.py (second half of the file = the last step)."""
import statistics as st, sys
rows=[l.split() for l in open(sys.argv[1] if len(sys.argv)>1 else '/root/repo/gpurun_out/attn_trace.txt')]
half=rows[len(rows)//2:]
f=[r for r in half if r[0][0]=='f']; b=[r for r in half if r[0][0]=='b']
print('fwd total ms', sum(int(x[1]) for x in f)/1e6, 'bwd total ms', sum(int(x[1]) for x in b)/1e6)
for t in (0,10,31,32,63,64,100,127,128,138,139):
    seg=f[t*24:(t+1)*24]
    self_=[int(x[1]) for x in seg[0::2]]; ctx=[int(x[1]) for x in seg[1::2]]
    print(t, seg[0][0], seg[1][0], 'self %.1f ctx %.1f'%(st.mean(self_)/1e3, st.mean(ctx)/1e3), 'grid', seg[0][2], seg[1][2])
print('bwd')
for t in (0,1,2,10,40,64,100,128,139):
    seg=b[t*24:(t+1)*24]
    a=[int(x[1]) for x in seg[0::2]]; c=[int(x[1]) for x in seg[1::2]]
    print(t, seg[0][0], seg[1][0], '%.1f %.1f'%(st.mean(a)/1e3, st.mean(c)/1e3), seg[0][2], seg[1][2])
