import itertools
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 += [[l+32 for l in g] for g in G128]
PLANE = 288//4  # plane stride in 4-dword groups = 72 -> mod 16 = 8
def phys(t, comp, s):
    # s: dict (comp, b1,b2,b3) -> 0/1 ; swaps halves of the 8-dword group
    return t ^ s[(comp, (t>>1)&1, (t>>2)&1, (t>>3)&1)]
def ok_reads_mid(s):
    for c in range(16):
        for g in G128:
            cols=set()
            for lane in g:
                j, comp = lane>>1, lane&1
                t = j + c + 64  # arbitrary aligned base (multiple of 2 groups)
                col = (phys(t, comp, s) + comp*PLANE) % 16
                if col in cols: return False
                cols.add(col)
    return True
def ok_reads_last(s):
    for comp in (0,1):
        for c in range(16):
            for g in G128:
                cols=set()
                for lane in g:
                    t = lane + c + 64
                    col = (phys(t, comp, s) + comp*PLANE) % 16
                    if col in cols: return False
                    cols.add(col)
    return True
def ok_writes(s):
    # two instructions: half h of 8-dword group 'j' (t = 2j + h) from lane (j, comp); 8-lane contiguous groups, bank mod 32 -> column mod 8
    for h in (0,1):
        for g in range(8):
            cols=set()
            for lane in range(8*g, 8*g+8):
                j, comp = lane>>1, lane&1
                t = 2*j + h + 64
                col = (phys(t, comp, s) + comp*PLANE) % 8
                if col in cols: return False
                cols.add(col)
    return True
keys = [(c,b1,b2,b3) for c in (0,1) for b1 in (0,1) for b2 in (0,1) for b3 in (0,1)]
found=[]
for bits in itertools.product((0,1), repeat=16):
    s = dict(zip(keys,bits))
    if ok_writes(s) and ok_reads_mid(s) and ok_reads_last(s):
        found.append(bits)
print(len(found))
for b in found[:10]: print(b)
# baseline checks
s0 = dict(zip(keys,[0]*16)); print("noswz", ok_writes(s0), ok_reads_mid(s0), ok_reads_last(s0))
s1 = {k:(1 if k[0] else 0) for k in keys}; print("comp swz", ok_writes(s1), ok_reads_mid(s1), ok_reads_last(s1))
print("all solutions as truth tables over (b1,b2,b3) for comp0 | comp1:")
import itertools
for b in found:
    # try to express as XOR of subset of variables + const
    s = dict(zip(keys,b))
    expr=None
    for const in (0,1):
        for mask in itertools.product((0,1),repeat=4):
            if all(s[k] == (const ^ (mask[0]&k[0]) ^ (mask[1]&k[1]) ^ (mask[2]&k[2]) ^ (mask[3]&k[3])) for k in keys):
                expr=(const,mask)
    print(b, "affine:", expr)
