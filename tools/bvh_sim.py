"""Offline comparison of hierarchy builders on config 5's mesh (no GPU): node visits, leaf pre-tests and leaves per walked ray
for the median split (rounds 1-3), a full-sweep surface-area split and the binned, depth-capped one the shim builds since
round 4 (rt_hip_shim.hip, BvhBuild::sah_split), under the kernels' traversal policy (nearer child first, boxes pruned by the
closest hit so far).  The median tree's figures agree with the PT_DIAG counts of the device walk (15.0 visits, 21.5 pre-tests).
usage: python tools/bvh_sim.py [rays]"""
import sys, numpy as np, ctypes as C
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'raytracer.c_amd'))
from rt_amd import scene as S, abi
sc = S.build_scene(5)
m = sc.meshes[0]
nt = m.mesh.num_triangles
V = np.ctypeslib.as_array(C.cast(m.mesh.vertices, C.POINTER(C.c_double)), shape=(nt*3, 5))[:, :3].reshape(nt, 3, 3).copy()
print('triangles', nt, 'bounds', V.reshape(-1,3).min(0), V.reshape(-1,3).max(0))
lo = V.min(1); hi = V.max(1); cen = V.mean(1)
LEAF = 15
def area(l, h):
    d = np.maximum(h - l, 0); return 2*(d[0]*d[1] + d[1]*d[2] + d[2]*d[0])
class Tree: pass
def build(kind, MAXD=99):
    nodes = []  # (box0, box1, ref0, ref1) ; ref = ('L', begin, end) or ('N', idx)
    order = np.arange(nt)
    depth = [0]
    def rec(b, e, level):
        idx = order[b:e]
        box = (lo[idx].min(0), hi[idx].max(0))
        if e - b <= LEAF:
            return ('L', b, e), box
        depth[0] = max(depth[0], level + 1)
        me = len(nodes); nodes.append(None)
        c = cen[idx]
        ext = c.max(0) - c.min(0)
        if kind == 'median':
            ax = int(np.argmax(ext)); mid = (e - b)//2
            o = np.argsort(c[:, ax], kind='stable'); order[b:e] = idx[o]
        elif kind == 'bin':
            n = e - b; NB = 64
            cap = LEAF * 2**(MAXD - level - 1)
            best = None
            cmin = c.min(0)
            for ax in range(3):
                if ext[ax] <= 0: continue
                bi = np.minimum(((c[:, ax] - cmin[ax]) * (NB / ext[ax])).astype(int), NB - 1)
                cnt = np.bincount(bi, minlength=NB)
                blo = np.full((NB, 3), 1e300); bhi = np.full((NB, 3), -1e300)
                np.minimum.at(blo, bi, lo[idx]); np.maximum.at(bhi, bi, hi[idx])
                l_lo = np.minimum.accumulate(blo, 0); l_hi = np.maximum.accumulate(bhi, 0)
                r_lo = np.minimum.accumulate(blo[::-1], 0)[::-1]; r_hi = np.maximum.accumulate(bhi[::-1], 0)[::-1]
                cl = np.cumsum(cnt)
                for sp in range(1, NB):
                    k = cl[sp - 1]
                    if k == 0 or k == n or k > cap or n - k > cap: continue
                    dl = l_hi[sp-1] - l_lo[sp-1]; dr = r_hi[sp] - r_lo[sp]
                    cost = 2*(dl[0]*dl[1]+dl[1]*dl[2]+dl[2]*dl[0])*k + 2*(dr[0]*dr[1]+dr[1]*dr[2]+dr[2]*dr[0])*(n-k)
                    if best is None or cost < best[0]: best = (cost, ax, sp, bi.copy(), k)
            if best is None:
                ax = int(np.argmax(ext)); mid = (e - b)//2
                o = np.argsort(c[:, ax], kind='stable'); order[b:e] = idx[o]
            else:
                _, ax, sp, bi, mid = best
                left = idx[bi < sp]; right = idx[bi >= sp]
                order[b:e] = np.concatenate([left, right])
        else:
            best = None
            n = e - b
            for ax in range(3):
                o = np.argsort(c[:, ax], kind='stable'); ii = idx[o]
                l_lo = np.minimum.accumulate(lo[ii], 0); l_hi = np.maximum.accumulate(hi[ii], 0)
                r_lo = np.minimum.accumulate(lo[ii][::-1], 0)[::-1]; r_hi = np.maximum.accumulate(hi[ii][::-1], 0)[::-1]
                dl = l_hi - l_lo; dr = r_hi - r_lo
                al = 2*(dl[:,0]*dl[:,1] + dl[:,1]*dl[:,2] + dl[:,2]*dl[:,0])
                ar = 2*(dr[:,0]*dr[:,1] + dr[:,1]*dr[:,2] + dr[:,2]*dr[:,0])
                k = np.arange(1, n)  # left gets k
                cost = al[k-1]*k + ar[k]*(n-k)
                cap = LEAF * 2**(MAXD - level - 1)
                cost = np.where((k <= cap) & (n - k <= cap), cost, np.inf)
                if kind == 'sah15':   # restrict splits to multiples that keep leaves fillable? plain
                    pass
                j = int(np.argmin(cost))
                if best is None or cost[j] < best[0]:
                    best = (cost[j], ax, k[j], ii)
            _, ax, mid, ii = best
            order[b:e] = ii
        r0, b0 = rec(b, b + mid, level + 1)
        r1, b1 = rec(b + mid, e, level + 1)
        nodes[me] = (b0, b1, r0, r1)
        return ('N', me), box
    root, box = rec(0, nt, 0)
    T = Tree(); T.nodes = nodes; T.order = order.copy(); T.root = root; T.depth = depth[0]; T.box = box
    return T
def slab(box, o, inv, tmax):
    t0 = (box[0] - o)*inv; t1 = (box[1] - o)*inv
    tn = np.minimum(t0, t1).max(); tf = np.maximum(t0, t1).min()
    tn = max(tn, 0.0)
    return (tn <= tf and tn < tmax), tn
def tri_hit(t, o, d):
    v0 = V[t,0]; e1 = V[t,1]-v0; e2 = V[t,2]-v0
    p = np.cross(d, e2); det = e1.dot(p)
    if abs(det) < 1e-12: return None
    inv = 1/det; s = o - v0; u = s.dot(p)*inv
    if u < 0 or u > 1: return None
    q = np.cross(s, e1); v = d.dot(q)*inv
    if v < 0 or u+v > 1: return None
    tt = e2.dot(q)*inv
    return tt if tt > 1e-8 else None
def traverse(T, o, d):
    inv = 1.0/np.where(d == 0, 1e-300, d)
    visits = 0; pre = 0; leaves = 0; tmin = np.inf
    stack = [T.root]
    maxsp = 0
    while stack:
        maxsp = max(maxsp, len(stack))
        ref = stack.pop()
        if ref[0] == 'L':
            leaves += 1
            for t in T.order[ref[1]:ref[2]]:
                pre += 1
                h = tri_hit(t, o, d)
                if h is not None and h < tmin: tmin = h
            continue
        visits += 1
        b0, b1, r0, r1 = T.nodes[ref[1]]
        h0, t0 = slab(b0, o, inv, tmin); h1, t1 = slab(b1, o, inv, tmin)
        if h0 and h1:
            if t0 <= t1: stack.append(r1); stack.append(r0)
            else: stack.append(r0); stack.append(r1)
        elif h0: stack.append(r0)
        elif h1: stack.append(r1)
    return visits, pre, leaves, tmin, maxsp
rng = np.random.default_rng(1)
c = (V.reshape(-1,3).min(0) + V.reshape(-1,3).max(0))/2
R = np.linalg.norm(V.reshape(-1,3) - c, axis=1).max()
def rays(n):
    out = []
    for _ in range(n):
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        o = c + a*R*rng.uniform(1.5, 6.0)
        while True:
            p = rng.uniform(-1, 1, 3)
            if p.dot(p) <= 1: break
        tgt = c + p*R
        d = tgt - o; d /= np.linalg.norm(d)
        out.append((o, d))
    return out
RS = rays(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
for kind, D in (('median', 99), ('sah', 10), ('bin', 10), ('bin', 11)):
    T = build(kind, D)
    tot = np.zeros(3); hits = 0; msp = 0
    for o, d in RS:
        v, p, l, t, sp = traverse(T, o, d)
        tot += (v, p, l); hits += np.isfinite(t); msp = max(msp, sp)
    n = len(RS)
    print(kind, D, 'nodes', len(T.nodes), 'depth', T.depth, 'per ray: visits %.2f pretests %.2f leaves %.2f' % tuple(tot/n), 'hit frac %.2f' % (hits/n), 'max stack', msp)
