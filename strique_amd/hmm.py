"""Profile-HMM topologies of the repeat counter and their 'baked' array form.

Host-side mirror of the reference's HMM classes (scripts/STRique.py:201-500): `profileHMM`,
`repeatHMM`, `flankedRepeatHMM`, `repeatModHMM`.  The reference builds pomegranate objects; here
the same states / transitions are written into a plain `Graph`, and `bake()` restates what
pomegranate 0.10.0's `HiddenMarkovModel.bake(merge='All')` (requirements.txt:11, called at
STRique.py:431,490) does to such a graph before `viterbi()` can run on it:

  1. states without in-edges (except start) or without out-edges (except end) are dropped,
     repeatedly -- this removes the unused start/end nodes of every embedded sub-model;
  2. every state whose out-edge probabilities do not sum to 1 (rounded to 8 decimals) is
     re-weighted in log space: logp -= log(round(sum, 8));
  3. silent states with a single certain (p == 1) out-edge are spliced out (also in front of the
     model end: adding log(1) = 0.0 is exact, so no path value changes);
  4. emitting states first, sorted by name; silent states after them in topological order.

The arrays go to the GPU through the C ABI (strq_model_create) and to the CPU oracle in tests.
pomegranate itself is not available in this image: SURVEY.md A.4 marks these semantics
"[recalled]"; the pre-bake topology is pinned by tests/golden/hmm_topology.json, which was
recorded from the reference's own classes.
"""
import math
from collections import namedtuple

import numpy as np

SQRT_2_PI = 2.50662827463      # the constant pomegranate's NormalDistribution uses

SILENT, NORMAL, UNIFORM = 0, 1, 2


class Graph(object):
    """States and transitions, nothing else."""

    def __init__(self):
        self.names, self.kinds, self.params = [], [], []
        self.edges = []                       # (src, dst, probability)
        self.layout = {}                      # optional: state -> (slot, lane) placement hint for the GPU kernel
        self.positions = {}                   # optional: emitting state -> (0 match-type | 1 insert-type, position along the profile chain)
        self.start = self.add_state("start", SILENT)
        self.end = self.add_state("end", SILENT)

    def add_state(self, name, kind, params=()):
        self.names.append(name); self.kinds.append(kind); self.params.append(tuple(float(p) for p in params))
        return len(self.names) - 1

    def add_transition(self, a, b, probability):
        self.edges.append((a, b, float(probability)))


PROFILE_DEFAULTS = {            # STRique.py:214-227
    'match_loop': .75, 'match_match': .15, 'match_insert': .09, 'match_delete': .01,
    'insert_loop': .15, 'insert_match_0': .40, 'insert_match_1': .40, 'insert_delete': .05,
    'delete_delete': .005, 'delete_insert': .05, 'delete_match': .945,
}

Profile = namedtuple("Profile", "s1 s2 e1 e2 match insert delete")


def _layer(defaults, overrides):
    tp = dict(defaults)
    if overrides:
        tp.update(overrides)
    return tp


def add_profile(g, sequence, pm, transition_probs=None, state_prefix='', no_silent=False,
                std_scale=1.0, std_offset=0.0):
    """One match/insert/delete column per k-mer of `sequence` (STRique.py:201-300)."""
    tp = _layer(PROFILE_DEFAULTS, transition_probs)
    k = pm.kmer
    n = len(sequence) - k + 1
    digits = int(np.ceil(np.log10(n)))
    match, insert, delete = [], [], []
    for idx in range(n):
        name = state_prefix + str(idx).rjust(digits, '0')
        mean, stdv = pm.model_dict[sequence[idx:idx + k]]
        match.append(g.add_state(name + 'm', NORMAL, (mean, stdv * std_scale + std_offset)))
        if not no_silent:
            delete.append(g.add_state(name + 'd', SILENT))
        insert.append(g.add_state(name + 'i', UNIFORM, (pm.model_min, pm.model_max)))
    s1 = g.add_state(state_prefix + 's1', SILENT); s2 = g.add_state(state_prefix + 's2', SILENT)
    e1 = g.add_state(state_prefix + 'e1', SILENT); e2 = g.add_state(state_prefix + 'e2', SILENT)
    last = n - 1
    for i in range(n):
        g.add_transition(match[i], match[i], tp['match_loop'])
        if i < last:
            g.add_transition(match[i], match[i + 1], tp['match_match'])
    for i in range(n):
        g.add_transition(insert[i], insert[i], tp['insert_loop'])
        g.add_transition(match[i], insert[i], tp['match_insert'])
        g.add_transition(insert[i], match[i], tp['insert_match_1'])
        if not no_silent and i < last:
            g.add_transition(insert[i], delete[i + 1], tp['insert_delete'])
        if i < last:
            g.add_transition(insert[i], match[i + 1], tp['insert_match_0'])
    if not no_silent:
        for i in range(n):
            g.add_transition(delete[i], insert[i], tp['delete_insert'])
            if i > 0:
                g.add_transition(match[i - 1], delete[i], tp['match_delete'])
            if i < last:
                g.add_transition(delete[i], match[i + 1], tp['delete_match'])
                g.add_transition(delete[i], delete[i + 1], tp['delete_delete'])
        g.add_transition(s1, delete[0], 1)
        g.add_transition(s2, match[0], 1)
        g.add_transition(delete[last], e1, tp['delete_delete'])
        g.add_transition(delete[last], e2, tp['delete_match'])
    else:
        for i in range(n - 2):
            g.add_transition(match[i], match[i + 2], tp['match_delete'])   # skip edge instead of a delete state
        g.add_transition(s1, insert[0], 1)
        g.add_transition(s2, match[0], 1)
    g.add_transition(insert[last], e1, tp['insert_delete'])
    g.add_transition(insert[last], e2, tp['insert_match_0'])
    g.add_transition(match[last], e2, tp['match_match'])
    g.add_transition(match[last], e1, tp['match_delete'])
    return Profile(s1, s2, e1, e2, match, insert, delete)


def extend_repeat(repeat, kmer):
    """Repeat unit padded so that it yields every k-mer of the tandem array, and the number of
    whole extra units that padding contains (STRique.py:328-335)."""
    if len(repeat) >= kmer:
        return repeat + repeat[:kmer - 1], 0
    ext = kmer - 1 + (len(repeat) - 1) - ((kmer - 1) % len(repeat))
    seq = repeat + (repeat * kmer)[:ext]
    return seq, int(len(seq) / len(repeat)) - 1


Repeat = namedtuple("Repeat", "s1 s2 e1 e2 d1 d2 profile repeat_offset")


def add_repeat(g, repeat, pm, transition_probs=None, state_prefix='', std_scale=1.0, std_offset=0.0):
    """One repeat unit closed into a loop through two emitting 'dummy' states (STRique.py:313-354)."""
    tp = _layer({'skip': .999, 'leave_repeat': .002}, transition_probs)
    seq, repeat_offset = extend_repeat(repeat, pm.kmer)
    prof = add_profile(g, seq, pm, tp, state_prefix, no_silent=True, std_scale=std_scale, std_offset=std_offset)
    d1 = g.add_state(state_prefix + 'dummy1', UNIFORM, (pm.model_min, pm.model_max))
    d2 = g.add_state(state_prefix + 'dummy2', UNIFORM, (pm.model_min, pm.model_max))
    e1 = g.add_state(state_prefix + 'e1', SILENT)
    e2 = g.add_state(state_prefix + 'e2', SILENT)
    g.add_transition(prof.e1, d1, 1)
    g.add_transition(prof.e2, d2, 1)
    g.add_transition(d1, e1, tp['leave_repeat'])
    g.add_transition(d2, e2, tp['leave_repeat'])
    g.add_transition(d1, prof.s1, 1 - tp['leave_repeat'])
    g.add_transition(d2, prof.s2, 1 - tp['leave_repeat'])
    return Repeat(prof.s1, prof.s2, e1, e2, d1, d2, prof, repeat_offset)


class FlankedRepeatModel(object):
    """prefix profile -> repeat loop -> suffix profile (STRique.py:384-431)."""

    def __init__(self, repeat, prefix, suffix, pm, config=None):
        tp = _layer({'skip': 1 - 1e-4, 'seq_std_scale': 1.0, 'rep_std_scale': 1.0,
                     'seq_std_offset': 0.0, 'rep_std_offset': 0.0, 'e1_ratio': 0.1},
                    config if isinstance(config, dict) else None)
        units = int(np.ceil(pm.kmer / len(repeat)))
        prefix_seq = prefix + (repeat * units)[:-1]
        suffix_seq = repeat * units + suffix
        self.flanking_count = units * 2 - 1
        g = Graph()
        pre = add_profile(g, prefix_seq, pm, tp, 'prefix', std_scale=tp['seq_std_scale'], std_offset=tp['seq_std_offset'])
        suf = add_profile(g, suffix_seq, pm, tp, 'suffix', std_scale=tp['seq_std_scale'], std_offset=tp['seq_std_offset'])
        rep = add_repeat(g, repeat, pm, tp, 'repeat', std_scale=tp['rep_std_scale'], std_offset=tp['rep_std_offset'])
        g.add_transition(g.start, pre.s1, tp['e1_ratio'])
        g.add_transition(g.start, pre.s2, 1 - tp['e1_ratio'])
        g.add_transition(pre.e1, rep.s1, 1)
        g.add_transition(pre.e2, rep.s2, 1)
        g.add_transition(rep.e1, suf.s1, 1)
        g.add_transition(rep.e2, suf.s2, 1)
        g.add_transition(suf.e1, g.end, 1)
        g.add_transition(suf.e2, g.end, 1)
        # placement hint for the Viterbi kernel: one slot per column type, lane = profile position, the
        # repeat unit right behind the prefix (its natural neighbour)
        # busy states (matches, the repeat unit) in the first two slots, plain inserts in the last two:
        # the kernel keeps 6 in-edge registers for the former and 3 for the latter
        lay = {}
        for p_, st in enumerate(pre.match): lay[st] = (0, p_)
        for p_, st in enumerate(suf.match): lay[st] = (1, p_)
        for p_, st in enumerate(pre.insert): lay[st] = (2, p_)
        for p_, st in enumerate(suf.insert): lay[st] = (3, p_)
        b0, b1 = len(pre.match), len(suf.match)
        for q_, st in enumerate(rep.profile.match): lay[st] = (0, b0 + q_)
        for q_, st in enumerate(rep.profile.insert): lay[st] = (1, b1 + q_)
        lay[rep.d2] = (0, b0 + len(rep.profile.match)); lay[rep.d1] = (1, b1 + len(rep.profile.insert))
        if max(l for _, l in lay.values()) < 64:
            g.layout = lay
        # the same states as one chain of positions: prefix profile, repeat unit, the two dummy states (match-type dummy2,
        # insert-type dummy1: they feed suffix00m / suffix00d the way a match / insert feeds the next position), suffix profile
        pos = {}
        P, R = len(pre.match), len(rep.profile.match)
        for p_, st in enumerate(pre.match): pos[st] = (0, p_)
        for p_, st in enumerate(pre.insert): pos[st] = (1, p_)
        for q_, st in enumerate(rep.profile.match): pos[st] = (0, P + q_)
        for q_, st in enumerate(rep.profile.insert): pos[st] = (1, P + q_)
        pos[rep.d2] = (0, P + R); pos[rep.d1] = (1, P + R)
        for p_, st in enumerate(suf.match): pos[st] = (0, P + R + 1 + p_)
        for p_, st in enumerate(suf.insert): pos[st] = (1, P + R + 1 + p_)
        g.positions = pos
        self.graph = g
        self.repeat_offset = rep.repeat_offset
        # visits of the two dummy states count repeat units (STRique.py:374-378)
        self.count_states = (rep.d1, rep.d2)
        self.count_bias = self.flanking_count - self.repeat_offset
        self.baked = bake(g, count_states=self.count_states, tag_substring='repeat')


class RepeatModModel(object):
    """Unmodified and modified repeat-unit profiles side by side between two emitting hub states
    (STRique.py:447-490)."""

    def __init__(self, repeat, pm_base, pm_mod, config=None):
        tp = _layer({'rep_std_scale': 1.5, 'rep_std_offset': 0.0, 'leave_repeat': .002},
                    config if isinstance(config, dict) else None)
        seq, _ = extend_repeat(repeat, pm_base.kmer)
        self.model_min = min(pm_base.model_min, pm_mod.model_min)
        self.model_max = max(pm_base.model_max, pm_mod.model_max)
        g = Graph()
        s0 = g.add_state('s0', UNIFORM, (self.model_min, self.model_max))
        e0 = g.add_state('e0', UNIFORM, (self.model_min, self.model_max))
        base = add_profile(g, seq, pm_base, tp, 'base', no_silent=True, std_scale=tp['rep_std_scale'],
                           std_offset=tp['rep_std_offset'])
        mod = add_profile(g, seq, pm_mod, tp, 'mod', no_silent=True,
                          std_scale=tp['rep_std_scale'] * pm_mod.scale2stdv(pm_base),
                          std_offset=tp['rep_std_offset'])
        g.add_transition(g.start, s0, 1)
        for p in (base, mod):
            g.add_transition(s0, p.s1, 0.25)
            g.add_transition(s0, p.s2, 0.25)
        for p in (base, mod):
            g.add_transition(p.e1, e0, 1)
            g.add_transition(p.e2, e0, 1)
        g.add_transition(e0, g.end, tp['leave_repeat'])
        g.add_transition(e0, s0, 1 - tp['leave_repeat'])
        lay = {}
        lane = 0
        for group in (base.match, base.insert, mod.match, mod.insert, [s0, e0]):
            for st in group:
                lay[st] = (0, lane); lane += 1
        if lane <= 64:
            g.layout = lay
        self.graph = g
        self.hub_states = (s0, e0)
        # tag 2 = hub states s0/e0 (group separators of mod_repeats, STRique.py:496), 1 = modified branch
        self.baked = bake(g, count_states=(), tag_substring='mod', tag2_states=(s0, e0))


# ---------------------------------------------------------------------------------------------
BakedHMM = namedtuple("BakedHMM", [
    "n_states", "silent_start", "start", "end",
    "in_ptr", "in_src", "in_logp",          # CSR of in-edges, sources ascending inside a state
    "emis_kind", "emis_a", "emis_b", "emis_c",
    "count_inc",                            # 1 for states whose visits are counted
    "tag",                                  # 1 for states whose name contains `tag_substring`
    "names", "orig_index",
    "hint_slot", "hint_lane",               # placement hints (-1 = none), see strq_model_create
    "pos_kind", "pos_index",                # position of every emitting state along the profile chain (None = unknown), see strq_model_set_positions
], defaults=(None, None))


def bake(g, count_states=(), tag_substring=None, tag2_states=()):
    n = len(g.names)
    alive = [True] * n
    edges = [(a, b, math.log(p) if p > 0 else -math.inf) for a, b, p in g.edges]
    # 1. orphans
    while True:
        indeg, outdeg = [0] * n, [0] * n
        for a, b, _ in edges:
            outdeg[a] += 1; indeg[b] += 1
        dead = [i for i in range(n) if alive[i] and
                ((indeg[i] == 0 and i != g.start) or (outdeg[i] == 0 and i != g.end))]
        if not dead:
            break
        for i in dead:
            alive[i] = False
        edges = [e for e in edges if alive[e[0]] and alive[e[1]]]
    # 2. out-edge normalisation in log space
    out = {}
    for idx, (a, b, lp) in enumerate(edges):
        out.setdefault(a, []).append(idx)
    for a, idxs in out.items():
        total = round(sum(math.e ** edges[i][2] for i in idxs), 8)
        if total != 1.0 and a != g.end:
            lt = math.log(total)
            for i in idxs:
                edges[i] = (edges[i][0], edges[i][1], edges[i][2] - lt)
    # 3. splice silent states with one certain out-edge
    while True:
        merged = 0
        out = {}
        for a, b, lp in edges:
            out.setdefault(a, []).append((b, lp))
        for a in range(n):
            if not alive[a] or g.kinds[a] != SILENT or a == g.start or a not in out:
                continue
            if len(out[a]) == 1 and out[a][0][1] == 0.0 and out[a][0][0] != a:
                b = out[a][0][0]
                edges = [(x, b if y == a else y, lp) for x, y, lp in edges if x != a]
                alive[a] = False
                merged += 1
                break
        if not merged:
            break
    # parallel edges (two spliced paths between the same pair of states, e.g. insert -> e1 -> e0
    # and insert -> e2 -> e0 in the modification model): Viterbi takes the better one, so keep a
    # single edge with the larger probability.  (pomegranate's networkx DiGraph would keep
    # whichever of the two it happened to re-insert last -- an order that depends on id() hashes
    # and differs between runs of the reference; see DESIGN.md "unpinned semantics".)
    best = {}
    for a, b, lp in edges:
        if (a, b) not in best or lp > best[(a, b)]:
            best[(a, b)] = lp
    seen = set()
    dedup = []
    for a, b, lp in edges:
        if (a, b) not in seen:
            seen.add((a, b)); dedup.append((a, b, best[(a, b)]))
    edges = dedup
    # 4. ordering
    emitting = sorted((i for i in range(n) if alive[i] and g.kinds[i] != SILENT), key=lambda i: (g.names[i], i))
    silent = [i for i in range(n) if alive[i] and g.kinds[i] == SILENT]
    sil_set = set(silent)
    indeg = {i: 0 for i in silent}
    succ = {i: [] for i in silent}
    for a, b, _ in edges:
        if a in sil_set and b in sil_set:
            if a == b:
                raise ValueError("silent self loop")
            indeg[b] += 1; succ[a].append(b)
    ready = sorted((i for i in silent if indeg[i] == 0), key=lambda i: (g.names[i], i))
    order = []
    while ready:
        i = ready.pop(0)
        order.append(i)
        newly = []
        for b in succ[i]:
            indeg[b] -= 1
            if indeg[b] == 0:
                newly.append(b)
        ready = sorted(ready + newly, key=lambda i: (g.names[i], i))
    if len(order) != len(silent):
        raise ValueError("cycle of silent states")
    final = emitting + order
    new = {old: k for k, old in enumerate(final)}
    m = len(final)
    ins = [[] for _ in range(m)]
    for a, b, lp in edges:
        ins[new[b]].append((new[a], lp))
    in_ptr = np.zeros(m + 1, np.int32)
    in_src, in_logp = [], []
    for k in range(m):
        ins[k].sort(key=lambda t: t[0])
        in_ptr[k + 1] = in_ptr[k] + len(ins[k])
        in_src += [t[0] for t in ins[k]]
        in_logp += [t[1] for t in ins[k]]
    ne = len(emitting)
    kind = np.zeros(ne, np.int32); ea = np.zeros(ne); eb = np.zeros(ne); ec = np.zeros(ne)
    for k, old in enumerate(emitting):
        kind[k] = g.kinds[old]
        p = g.params[old]
        if g.kinds[old] == NORMAL:
            mu, sigma = p
            ea[k] = mu
            eb[k] = 1.0 / (2 * sigma ** 2)
            ec[k] = -math.log(sigma * SQRT_2_PI)
        else:
            lo, hi = p
            ea[k], eb[k] = lo, hi
            ec[k] = -math.log(hi - lo)
    count_inc = np.zeros(m, np.int32)
    for s in count_states:
        count_inc[new[s]] = 1
    tag = np.array([2 if old in tag2_states else (1 if (tag_substring and tag_substring in g.names[old]) else 0)
                    for old in final], np.int32)
    hint_slot = np.full(m, -1, np.int32); hint_lane = np.full(m, -1, np.int32)
    if g.layout and all(old in g.layout for old in emitting):
        for old in emitting:
            hint_slot[new[old]], hint_lane[new[old]] = g.layout[old]
    pos_kind = pos_index = None
    if getattr(g, 'positions', None) and all(old in g.positions for old in emitting):
        pos_kind = np.zeros(m, np.int32); pos_index = np.zeros(m, np.int32)
        for old in emitting:
            pos_kind[new[old]], pos_index[new[old]] = g.positions[old]
    return BakedHMM(m, ne, new[g.start], new[g.end], in_ptr, np.array(in_src, np.int32),
                    np.array(in_logp, np.float64), kind, ea, eb, ec, count_inc, tag,
                    [g.names[i] for i in final], np.array(final, np.int32), hint_slot, hint_lane, pos_kind, pos_index)
