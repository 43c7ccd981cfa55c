"""Search of a bank-conflict-free slot function for transposed bf16 images with FOUR 16-byte chunks (8 rows each) per column
(k_rot_l1_bwd_sp: 32-row half tiles): staging writes = ds_write_b128 from 8 contiguous lanes holding columns 4 apart, fragment
reads = ds_read_b128 in the four 16-lane groups of MI355X_MICROARCH.md over 32 consecutive columns.  tn4_slot is (2, -, 4, -)."""
import itertools
RG=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG=RG+[[x+32 for x in g] for g in RG]
def ok(slot):
    # reads: lane l: i=l&31,h=l>>5 ; for blk in 0..7, ks in 0..1: chunk=2ks+h ; ds_read_b128: 16 distinct slots mod 16 per group
    for blk in range(8):
        for ks in range(2):
            for g in RG:
                s=set()
                for l in g:
                    i=l&31;h=l>>5
                    s.add(slot(blk*32+i,2*ks+h)%16)
                if len(s)!=16: return False
    # writes: ds_write_b128: 8 contiguous lanes, slots mod 8 distinct; lane l -> columns 4l+q, chunk=wave
    for q in range(4):
        for w in range(4):
            for g0 in range(0,64,8):
                s=set(slot(4*l+q,w)%8 for l in range(g0,g0+8))
                if len(s)!=8: return False
    # bijection
    allv=set(slot(c,ch) for c in range(256) for ch in range(4))
    return len(allv)==1024 and max(allv)<1024
best=[]
shifts=[1,2,3,4,5,6]
for s1 in shifts+[None]:
  for s1b in shifts+[None]:
    for s2 in shifts+[None]:
      for s2b in shifts+[None]:
        def f(c,ch,s1=s1,s1b=s1b,s2=s2,s2b=s2b):
            a=0
            if s1 is not None: a^=(c>>s1)
            if s1b is not None: a^=(c>>s1b)
            b=0
            if s2 is not None: b^=(c>>s2)
            if s2b is not None: b^=(c>>s2b)
            return (c>>2)*16 + (((c&3)^(b&3))<<2) + ((ch^a)&3)
        if ok(f): best.append((s1,s1b,s2,s2b))
print(len(best), best[:10])
