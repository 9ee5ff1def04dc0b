"""tools/models/cut_tree_walk.py -- CPU model of the shadow tree cut below its top levels (restir_amd/csrc/scene.hip build_occlusion_side,
rs_scene.h walk_occlusion_tree<true>): LDS slots in pre-order, link records below the inner nodes of the last level and after the
last slot; random trees and random box-test outcomes, the cut walk must visit the nodes of the plain walk in the same order."""
import random, sys
sys.setrecursionlimit(100000)
def build(n_leaves, rng):
    nodes = []   # (prim, next)
    def rec(cnt):
        me = len(nodes); nodes.append(None)
        if cnt == 1 or (cnt <= 4 and rng.random() < 0.3):
            nodes[me] = [cnt, None]
        else:
            l = rng.randint(1, cnt - 1)
            rec(l); rec(cnt - l)
            nodes[me] = [-1, None]
        nodes[me][1] = len(nodes)
        return me
    rec(n_leaves)
    return nodes
def cut_tree(nodes, cap):
    no = len(nodes)
    depth = [0]*no; ends = []
    for i in range(no):
        while ends and ends[-1] <= i: ends.pop()
        depth[i] = len(ends); ends.append(nodes[i][1])
    D = -1
    for d in range(30):
        cnt = 1 + sum((2 if (depth[i] == d and nodes[i][0] < 0) else 1) for i in range(no) if depth[i] <= d)
        if cnt > cap: break
        D = d
    if D < 1: return None
    slot = [-1]*(no+1); ns = 0
    for i in range(no):
        if depth[i] <= D:
            slot[i] = ns; ns += 1
            if depth[i] == D and nodes[i][0] < 0: ns += 1
    def ref(t):
        if t >= no: return no*16
        return slot[t]*16+1 if slot[t] >= 0 else t*16
    ns += 1
    cutw = [None]*no; top = [None]*ns
    top[ns-1] = ('jump', no*16, None)
    for i in range(no):
        inner = nodes[i][0] < 0
        cutw[i] = ('inner', ref(nodes[i][1]), i) if inner else ('leaf', None, i)
        if slot[i] < 0: continue
        top[slot[i]] = cutw[i]
        if inner and depth[i] == D:
            top[slot[i]+1] = ('jump', (i+1)*16, None)
    return top, cutw, D
def walk_plain(nodes, outcome):
    cur = 0; seq = []
    while cur != len(nodes):
        seq.append(cur)
        leaf = nodes[cur][0] >= 0
        if outcome(cur) or leaf: cur += 1
        else: cur = nodes[cur][1]
    return seq
def walk_cut(nodes, top, cutw, outcome):
    end = len(nodes)*16; cur = 1; seq = []; steps = 0
    while cur != end:
        steps += 1
        assert steps < 10*len(nodes), "no termination"
        rec = top[cur >> 4] if cur & 1 else cutw[cur >> 4]
        kind, link, gid = rec
        if kind == 'jump': cur = link; continue
        seq.append(gid)
        if kind == 'leaf' or outcome(gid): cur += 16
        else: cur = link
    return seq
rng = random.Random(1)
for trial in range(300):
    n = rng.randint(2, 3000)
    nodes = build(n, rng)
    cap = rng.choice([8, 16, 64, 768])
    r = cut_tree(nodes, cap)
    if r is None: continue
    top, cutw, D = r
    assert len(top) <= cap
    for w in range(20):
        p = rng.random()
        table = [rng.random() < p for _ in nodes]
        a = walk_plain(nodes, lambda i: table[i])
        b = walk_cut(nodes, top, cutw, lambda i: table[i])
        assert a == b, (trial, w)
print("cut-tree model ok")
