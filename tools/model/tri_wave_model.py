"""A scheduling model of one wave of k_trace_lean_triangles (config 5): replays the per-ray event traces of a sample of 8x8 tiles
(tools/model/tri_events.c, the oracle's traversal) under the kernel's present policy and under candidate policies, and counts what
the wave would issue: node-loop bodies, triangle rounds, chunks of 64 (leaf, triangle) pairs.  Study aid; runs on the CPU box.
    gcc -O2 -ffp-contract=off -shared -fPIC -fopenmp -o build/libtri_events.so tools/model/tri_events.c -lm
    python tools/model/tri_wave_model.py [tiles [dim W H]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import orc

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dim, W, H = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 3840, 2160)
CAP = 256
cache = os.path.join(ROOT, "build", f"tri_events_{dim}_{W}x{H}_{NT}.npz")
if os.path.exists(cache):
    z = np.load(cache)
    ev, cnt = z["ev"], z["cnt"]
else:
    g = orc.test_sphere_grid(dim)
    nodes = orc.build_flat_octree(g)
    tris, off = orc.build_leaf_triangles(g, nodes)
    cam = orc.Camera(0.5, 0.7, 1.8)
    view, pos = cam.get_view(), cam.get_pos()
    # live tiles: inside the sphere's silhouette (and its rim): sample tiles within the projected disc, found from a coarse frame
    small, _ = orc.render(nodes, g.min, g.voxel_size, view, pos, W / H, 45.0, W // 8, H // 8, nthreads=8)
    live = np.argwhere(small[..., 3] * (small[..., 0] > 0) > 0)            # (ty, tx) of coarse pixels that hit
    rng = np.random.default_rng(5)
    pick = live[rng.choice(len(live), size=min(NT, len(live)), replace=False)]
    tiles = np.ascontiguousarray(pick[:, ::-1], np.int32)                   # (tx, ty)
    L = C.CDLL(os.path.join(ROOT, "build", "libtri_events.so"))
    ev = np.zeros((len(tiles), 64, 2, CAP), np.int16)
    cnt = np.zeros((len(tiles), 64, 2), np.int32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.c_void_p)
    L.tri_events_tiles(C.c_void_p(nodes.ctypes.data), C.c_void_p(tris.ctypes.data), C.c_void_p(off.ctypes.data), f32(g.min), C.c_float(g.voxel_size),
                       f32(view), f32(pos), C.c_float(W / H), C.c_float(45.0), W, H, C.c_void_p(tiles.ctypes.data), len(tiles), CAP,
                       C.c_void_p(ev.ctypes.data), C.c_void_p(cnt.ctypes.data))
    np.savez_compressed(cache, ev=ev, cnt=cnt)
assert cnt.max() < CAP
tight = ev >= 1000                      # tri_events.c: a missed leaf test that a box around the leaf's own triangles (+1e-3 voxel) would have rejected
ev = np.where(tight, ev - 1000, ev).astype(np.int16)
print(f"{len(ev)} tiles; per pixel: primary events {cnt[..., 0].mean():.1f}, shadow events {cnt[..., 1].mean():.1f}")


DROP_TIGHT = False                      # True: the leaf tests a tight box would reject vanish from the streams (free, immediate rejection)


def rays_of_tile(t):
    """per lane: list of rays (primary[, shadow]); a ray = list of events"""
    out = []
    for lane in range(64):
        rs = []
        for k in range(2):
            n = cnt[t, lane, k]
            if n:
                r = ev[t, lane, k, :n]
                rs.append((r[~tight[t, lane, k, :n]] if DROP_TIGHT else r).tolist())
        out.append(rs)
    return out


class Lane:
    __slots__ = ("rays", "ri", "pos", "pend", "blocked", "needpop")

    def __init__(self, rays):
        self.rays, self.ri, self.pos = rays, 0, 0
        self.pend = []          # speculation: [(cnt, hit)] leaf tests popped and not resolved yet
        self.blocked = False
        self.needpop = False    # speculation: the lane's next body only pops its stack (no visit)

    def cur(self):
        if self.ri >= len(self.rays):
            return None
        r = self.rays[self.ri]
        return r[self.pos] if self.pos < len(r) else "end"

    def next_ray(self):
        self.ri += 1
        self.pos = 0
        self.pend = []
        self.blocked = False
        self.needpop = False


def policy_current(rays, batch=16):
    lanes = [Lane(r) for r in rays]
    bodies = rounds = chunks = pairs = lane_trips = 0
    while True:
        # lanes whose ray has run out end it at the round's end (the "ended" block); model: at the top of a round
        waiting = sum(1 for l in lanes if isinstance(l.cur(), int) and l.cur() != 0)
        if waiting < batch:
            while True:
                alive = [l for l in lanes if l.cur() == 0]
                if not alive:
                    break
                bodies += 1
                lane_trips += len(alive)
                for l in alive:
                    l.pos += 1
                    c = l.cur()
                    if isinstance(c, int) and c != 0:
                        waiting += 1
                if waiting >= batch:
                    break
        rounds += 1
        tot = 0
        for l in lanes:
            c = l.cur()
            if isinstance(c, int) and c != 0:
                tot += abs(c)
                l.pos += 1
                if c < 0:
                    l.next_ray()            # hit: primary -> shadow starts; shadow -> done
        chunks += (tot + 63) // 64
        pairs += tot
        for l in lanes:
            if l.cur() == "end":
                l.next_ray()                # the ray missed everything
        if all(l.cur() is None for l in lanes):
            break
    return dict(bodies=bodies, rounds=rounds, chunks=chunks, pairs=pairs, lane_trips=lane_trips)


def policy_speculate(rays, batch=16, K=2, heads_only=True, full_chunks=False):
    """A lane that pops a triangle leaf files it (at most K unresolved) and walks on as if the test missed; its next body only pops the
    stack.  A round resolves the heads (or all) of the files: a hit ends the ray there (what it walked beyond is dropped)."""
    lanes = [Lane(r) for r in rays]
    bodies = rounds = chunks = pairs = lane_trips = wasted = 0

    def can_walk(l):
        if l.cur() is None or l.blocked:
            return False
        if l.pend and any(h for _, h in l.pend) and False:
            return False
        c = l.cur()
        return l.needpop or c == 0

    while True:
        filed = sum(1 for l in lanes if l.pend)
        if filed < batch:
            while True:
                alive = [l for l in lanes if can_walk(l)]
                if not alive:
                    break
                bodies += 1
                lane_trips += len(alive)
                for l in alive:
                    doomed = any(h for _, h in l.pend)
                    if doomed:
                        wasted += 1
                    if l.needpop:
                        l.needpop = False           # this body popped the stack; what came out is l.cur()
                    else:
                        l.pos += 1                  # visited a node; the body also descended into the next candidate
                    c = l.cur()
                    while isinstance(c, int) and c != 0:        # the child popped is a triangle leaf
                        if len(l.pend) >= K:
                            l.blocked = True
                            break
                        if not l.pend:
                            filed += 1
                        l.pend.append((abs(c), c < 0))
                        l.pos += 1
                        l.needpop = True
                        break
                if filed >= batch:
                    break
        rounds += 1
        tot = 0
        for l in lanes:
            if not l.pend:
                continue
            todo = l.pend[:1] if heads_only else l.pend
            hit = False
            ndone = 0
            for c, h in todo:
                tot += c
                ndone += 1
                if h:
                    hit = True
                    break
            if hit:
                l.next_ray()
            else:
                l.pend = l.pend[ndone:]
                if l.blocked:
                    l.blocked = False
                    # the leaf it stands on is filed now
                    c = l.cur()
                    l.pend.append((abs(c), c < 0))
                    l.pos += 1
                    l.needpop = True
        chunks += (tot + 63) // 64
        pairs += tot
        for l in lanes:
            if l.cur() == "end" and not l.pend and not l.blocked:
                l.next_ray()
            elif l.cur() == "end" and l.needpop:
                l.needpop = False
        if all(l.cur() is None for l in lanes):
            break
    return dict(bodies=bodies, rounds=rounds, chunks=chunks, pairs=pairs, lane_trips=lane_trips, wasted=wasted)


def policy_carry(rays, batch=16, pair_threshold=0, full_only=True):
    """The present policy, but a round tests only FULL chunks of 64 pairs (the remainder's leaves keep waiting, with their progress kept)
    unless nobody can walk; pair_threshold > 0: the round starts when that many pairs wait (instead of `batch` lanes)."""
    lanes = [Lane(r) for r in rays]
    left = [0] * 64         # pairs of the waiting leaf not tested yet
    bodies = rounds = chunks = pairs = lane_trips = 0

    def is_leaf(c):
        return isinstance(c, int) and c != 0

    while True:
        for i, l in enumerate(lanes):
            if is_leaf(l.cur()) and left[i] == 0:
                left[i] = abs(l.cur())
        waiting = sum(1 for l in lanes if is_leaf(l.cur()))
        wpairs = sum(left)
        trig = (lambda: wpairs >= pair_threshold) if pair_threshold else (lambda: waiting >= batch)
        if not trig():
            while True:
                alive = [(i, l) for i, l in enumerate(lanes) if l.cur() == 0]
                if not alive:
                    break
                bodies += 1
                lane_trips += len(alive)
                for i, l in alive:
                    l.pos += 1
                    c = l.cur()
                    if is_leaf(c):
                        waiting += 1
                        left[i] = abs(c)
                        wpairs += abs(c)
                if trig():
                    break
        rounds += 1
        anyone_walks = any(l.cur() == 0 for l in lanes)
        tot = sum(left)
        budget = (tot // 64) * 64 if (full_only and anyone_walks and tot >= 64) else tot
        if full_only and anyone_walks and tot < 64:
            budget = tot                      # a round was asked for with less than a chunk: test it (the trigger decides how often)
        used = 0
        for i, l in enumerate(lanes):
            if left[i] == 0:
                continue
            take = min(left[i], budget - used)
            used += take
            left[i] -= take
            if left[i] == 0:
                c = l.cur()
                l.pos += 1
                if c < 0:
                    l.next_ray()
            if used >= budget:
                break
        chunks += (used + 63) // 64
        pairs += used
        for l in lanes:
            if l.cur() == "end":
                l.next_ray()
        if all(l.cur() is None for l in lanes):
            break
    return dict(bodies=bodies, rounds=rounds, chunks=chunks, pairs=pairs, lane_trips=lane_trips)


def policy_dual(raysA, raysB, batch=16, switch_in_round=True):
    """Two tiles per wave: lane l owns a ray context of each; a body serves the first of its contexts that can walk.  A round tests the
    waiting leaves of both sets."""
    A = [Lane(r) for r in raysA]
    B = [Lane(r) for r in raysB]
    bodies = rounds = chunks = pairs = lane_trips = 0
    is_leaf = lambda c: isinstance(c, int) and c != 0
    while True:
        waiting = sum(1 for l in A + B if is_leaf(l.cur()))
        chosen = [a if a.cur() == 0 else b for a, b in zip(A, B)]       # switch_in_round = False: a lane's context is fixed for the round
        if waiting < batch:
            while True:
                act = []
                if switch_in_round:
                    for a, b in zip(A, B):
                        if a.cur() == 0:
                            act.append(a)
                        elif b.cur() == 0:
                            act.append(b)
                else:
                    act = [c for c in chosen if c.cur() == 0]
                if not act:
                    break
                bodies += 1
                lane_trips += len(act)
                for l in act:
                    l.pos += 1
                    if is_leaf(l.cur()):
                        waiting += 1
                if waiting >= batch:
                    break
        rounds += 1
        tot = 0
        for l in A + B:
            c = l.cur()
            if is_leaf(c):
                tot += abs(c)
                l.pos += 1
                if c < 0:
                    l.next_ray()
        chunks += (tot + 63) // 64
        pairs += tot
        for l in A + B:
            if l.cur() == "end":
                l.next_ray()
        if all(l.cur() is None for l in A + B):
            break
    return dict(bodies=bodies, rounds=rounds, chunks=chunks, pairs=pairs, lane_trips=lane_trips)


def policy_dual_parked(raysA, raysB, batch=16):
    """Two tiles per wave, the simple form: a lane's second context is PARKED (registers swapped at a round's end when the active one
    cannot walk and the parked one can); a round tests the ACTIVE contexts' leaves only."""
    act = [Lane(r) for r in raysA]
    par = [Lane(r) for r in raysB]
    bodies = rounds = chunks = pairs = lane_trips = swaps = 0
    is_leaf = lambda c: isinstance(c, int) and c != 0
    while True:
        waiting = sum(1 for l in act if is_leaf(l.cur()))
        if waiting < batch:
            while True:
                alive = [l for l in act if l.cur() == 0]
                if not alive:
                    break
                bodies += 1
                lane_trips += len(alive)
                for l in alive:
                    l.pos += 1
                    if is_leaf(l.cur()):
                        waiting += 1
                if waiting >= batch:
                    break
        rounds += 1
        tot = 0
        for l in act:
            c = l.cur()
            if is_leaf(c):
                tot += abs(c)
                l.pos += 1
                if c < 0:
                    l.next_ray()
        chunks += (tot + 63) // 64
        pairs += tot
        for l in act:
            if l.cur() == "end":
                l.next_ray()
        sw = False
        for i in range(64):
            a, b = act[i], par[i]
            if a.cur() != 0 and (b.cur() == 0 or (a.cur() is None and b.cur() is not None)):
                act[i], par[i] = b, a
                sw = True
        swaps += 1 if sw else 0
        if all(l.cur() is None for l in act + par):
            break
    return dict(bodies=bodies, rounds=rounds, chunks=chunks, pairs=pairs, lane_trips=lane_trips, swaps=swaps)


def run_dual(name, **kw):
    tot = {}
    n = len(ev) // 2
    fn = kw.pop("fn", policy_dual)
    for t in range(n):
        r = fn(rays_of_tile(2 * t), rays_of_tile(2 * t + 1), **kw)
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    est = tot["bodies"] * 137 + tot["rounds"] * 110 + tot["chunks"] * 87
    print(f"{name:44s} PER TILE: bodies {tot['bodies'] / n / 2:6.1f} (util {tot['lane_trips'] / tot['bodies'] / 64:.2f})  rounds {tot['rounds'] / n / 2:5.1f}  chunks {tot['chunks'] / n / 2:5.1f} "
          f"(fill {tot['pairs'] / max(1, tot['chunks']) / 64:.2f})  est. instr/tile {est / n / 2:7.0f}")


def run(name, fn, **kw):
    tot = {}
    for t in range(len(ev)):
        r = fn(rays_of_tile(t), **kw)
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    n = len(ev)
    est = tot["bodies"] * 137 + tot["rounds"] * 110 + tot["chunks"] * 87
    print(f"{name:44s} bodies {tot['bodies'] / n:6.1f} (util {tot['lane_trips'] / tot['bodies'] / 64:.2f})  rounds {tot['rounds'] / n:5.1f}  chunks {tot['chunks'] / n:5.1f} "
          f"(fill {tot['pairs'] / max(1, tot['chunks']) / 64:.2f})  pairs {tot['pairs'] / n:6.0f}  est. instr/wave {est / n:7.0f}" + (f"  wasted lane-bodies {tot['wasted'] / n:.1f}" if "wasted" in tot else ""))
    return est / n


base = run("current (round at 16 waiting)", policy_current)
for b in (8, 12, 24):
    run(f"current, round at {b}", policy_current, batch=b)
for K in (1, 2, 3):
    for b in (16, 24, 32):
        run(f"speculate K={K}, round at {b} filed, heads", policy_speculate, batch=b, K=K)
for b in (12, 16, 24, 32, 48):
    run_dual(f"two tiles per wave, round at {b} waiting", batch=b)
for b in (12, 16, 24, 32):
    run_dual(f"two tiles, context switch at round ends only, {b}", batch=b, switch_in_round=False)
for b in (8, 12, 16, 24):
    run_dual(f"two tiles, second context parked, round at {b}", batch=b, fn=policy_dual_parked)
for b in (12, 16, 20):
    run(f"full chunks only, round at {b} lanes", policy_carry, batch=b)
for pt in (48, 64, 96, 128):
    run(f"round at {pt} waiting pairs, every pair tested", policy_carry, pair_threshold=pt, full_only=False)
    run(f"round at {pt} waiting pairs, full chunks only", policy_carry, pair_threshold=pt, full_only=True)
run("speculate K=2, round at 16, all filed tests", policy_speculate, batch=16, K=2, heads_only=False)
run("speculate K=3, round at 24, all filed tests", policy_speculate, batch=24, K=3, heads_only=False)

miss = ev > 0
print(f"missed leaf tests a tight box would reject: {tight.sum() / miss.sum():.2f} of them, {ev[tight].sum() / np.abs(ev).sum():.2f} of all pairs")
DROP_TIGHT = True
run("tight boxes, rejection free and immediate (bound)", policy_current)
DROP_TIGHT = False

# ---- where the idle lanes of the node loop come from
mx = tot = n = 0
for t in range(len(ev)):
    per_lane = [sum(1 for r in rs for e in r if e == 0) for rs in rays_of_tile(t)]
    mx += max(per_lane); tot += sum(per_lane); n += 1
print(f"per wave: lane-trips / 64 = {tot / n / 64:.1f} bodies (perfect packing); busiest lane's trips = {mx / n:.1f} (no waiting at all); the present policy issues more than either (first line)")
