"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The profile HMMs of STRique's repeat counter as the reference hands them to pomegranate, kept
UN-BAKED: every silent state (s1/s2/e1/e2 hubs, delete states) stays in the graph, nothing is
spliced, no two edges are ever merged.  `prepare()` only does what cannot be avoided before a
Viterbi decode can run on the graph:

    * orphan removal             (the unused start/end nodes of embedded sub-models),
    * out-edge renormalisation   (pomegranate 0.10.0 `bake`: states whose out-probabilities, rounded
                                  to 8 decimals, do not sum to 1 are re-weighted in log space),
    * an order                   (emitting states sorted by name, silent states topologically).

This module does not import anything from strique_amd: the product's `hmm.bake()` (splicing of
certain silent states, parallel-edge rule, its own ordering) is checked AGAINST it in
tests/test_oracle_independent.py -- same log-probability bits, same emitting path, same count.

Reference being restated (scripts/STRique.py):
    profileHMM            :201-307     repeatHMM          :313-378
    flankedRepeatHMM      :384-441     repeatModHMM       :447-500
pomegranate v0.10.0 (requirements.txt:10-11) is not in /root/reference and not installable here;
its graph container is networkx < 2.0 (requirements.txt:5): a node dict and per-node successor
dicts, insertion ordered on the Python >= 3.6 the reference runs on (Dockerfile: ubuntu 18.04,
.travis.yml: 3.6 / 3.7 / 3.8).  `Net` keeps exactly that order: sub-models are complete graphs of
their own (with their own start / end nodes) that are united into the parent, as
`HiddenMarkovModel.add_model` does with `networkx.union`.

PARITY STATUS: the pre-bake topology is pinned -- tests/golden/hmm_topology*.json were recorded from
the reference's own classes (tests/golden/make_golden.py) and `Net.from_recording` loads them
directly, so for the bundled targets the decode runs on the reference's graph, not on a
restatement of it.  The renormalisation rule and the emission formulas are "[recalled]"
pomegranate semantics -- unpinned.
"""
import math

import numpy as np

SILENT, NORMAL, UNIFORM = 0, 1, 2
_KIND = {"silent": SILENT, "normal": NORMAL, "uniform": UNIFORM}
SQRT_2_PI = 2.50662827463       # pomegranate's constant


class Net(object):
    """Directed graph with networkx-1.x semantics: adding an edge adds unknown end points, adding an
    existing edge again overwrites its probability, iteration follows insertion order."""

    def __init__(self, label="model"):
        self.name, self.kind, self.par = [], [], []
        self.succ = {}
        self.start = self.node(label + "-start", SILENT)
        self.end = self.node(label + "-end", SILENT)
        self.counted = ()            # nodes whose visits count repeat units

    def node(self, name, kind, par=()):
        self.name.append(name); self.kind.append(kind); self.par.append(tuple(float(x) for x in par))
        self.succ[len(self.name) - 1] = {}
        return len(self.name) - 1

    def edge(self, a, b, p):
        self.succ[a][b] = float(p)

    def edges(self):
        for a, nb in self.succ.items():
            for b, p in nb.items():
                yield a, b, p

    def unite(self, other):
        """networkx.union(self, other): other's nodes and edges are appended; returns the index map."""
        base = len(self.name)
        for i in range(len(other.name)):
            self.node(other.name[i], other.kind[i], other.par[i])
        for a, b, p in other.edges():
            self.edge(base + a, base + b, p)
        return lambda i: base + i

    @classmethod
    def from_recording(cls, t, counted_names=()):
        """A graph recorded from the reference's classes (tests/golden/hmm_topology*.json)."""
        g = cls.__new__(cls)
        g.name = [s["name"] for s in t["states"]]
        g.kind = [_KIND[s["kind"]] for s in t["states"]]
        g.par = [tuple(float(x) for x in s["params"]) for s in t["states"]]
        g.succ = {i: {} for i in range(len(g.name))}
        for a, b, p, _group in t["edges"]:
            g.edge(a, b, p)
        g.start, g.end = t["start"], t["end"]
        g.counted = tuple(i for i, n in enumerate(g.name) if n in counted_names and g.kind[i] != SILENT)
        return g


# ---------------------------------------------------------------------------------------------
# topology (STRique.py:201-500)
# ---------------------------------------------------------------------------------------------
_PROFILE_P = dict(match_loop=.75, match_match=.15, match_insert=.09, match_delete=.01,
                  insert_loop=.15, insert_match_0=.40, insert_match_1=.40, insert_delete=.05,
                  delete_delete=.005, delete_insert=.05, delete_match=.945)        # :214-227


def _merged(defaults, override):
    out = dict(defaults)
    if override:
        out.update(override)
    return out


class _Profile(object):
    """profileHMM: a model of its own (:201-300); hubs s1/s2 (in), e1/e2 (out)."""

    def __init__(self, sequence, pm, probs, tag, no_silent, std_scale, std_offset):
        P = _merged(_PROFILE_P, probs)
        net = Net(tag + "profile")
        K = pm.kmer
        kmers = [sequence[i:i + K] for i in range(len(sequence) - K + 1)]
        width = int(np.ceil(np.log10(len(kmers))))
        label = [tag + str(i).rjust(width, "0") for i in range(len(kmers))]
        M = [net.node(label[i] + "m", NORMAL, (pm.table[k][0], pm.table[k][1] * std_scale + std_offset))
             for i, k in enumerate(kmers)]
        I = [net.node(label[i] + "i", UNIFORM, (pm.model_min, pm.model_max)) for i in range(len(kmers))]
        D = [] if no_silent else [net.node(label[i] + "d", SILENT) for i in range(len(kmers))]
        # the reference adds the match states, then the insert states, then the delete states (:262-266);
        # node() above ran in list-comprehension order M, I, D which is the same
        s1, s2, e1, e2 = (net.node(tag + h, SILENT) for h in ("s1", "s2", "e1", "e2"))
        L = len(kmers) - 1
        for i in range(L + 1):                                   # matches (:268-271)
            net.edge(M[i], M[i], P["match_loop"])
            if i < L:
                net.edge(M[i], M[i + 1], P["match_match"])
        for i in range(L + 1):                                   # insertions (:273-280)
            net.edge(I[i], I[i], P["insert_loop"])
            net.edge(M[i], I[i], P["match_insert"])
            net.edge(I[i], M[i], P["insert_match_1"])
            if D and i < L:
                net.edge(I[i], D[i + 1], P["insert_delete"])
            if i < L:
                net.edge(I[i], M[i + 1], P["insert_match_0"])
        if D:                                                    # deletions (:282-295)
            for i in range(L + 1):
                net.edge(D[i], I[i], P["delete_insert"])
                if i > 0:
                    net.edge(M[i - 1], D[i], P["match_delete"])
                if i < L:
                    net.edge(D[i], M[i + 1], P["delete_match"])
                    net.edge(D[i], D[i + 1], P["delete_delete"])
            net.edge(s1, D[0], 1); net.edge(s2, M[0], 1)
            net.edge(D[L], e1, P["delete_delete"]); net.edge(D[L], e2, P["delete_match"])
        else:                                                    # skip edges instead (:296-301)
            for i in range(L - 1):
                net.edge(M[i], M[i + 2], P["match_delete"])
            net.edge(s1, I[0], 1); net.edge(s2, M[0], 1)
        net.edge(I[L], e1, P["insert_delete"]); net.edge(I[L], e2, P["insert_match_0"])
        net.edge(M[L], e2, P["match_match"]); net.edge(M[L], e1, P["match_delete"])
        self.net, self.s1, self.s2, self.e1, self.e2 = net, s1, s2, e1, e2


def tandem_unit(repeat, K):
    """The stretch whose k-mers are all k-mers of the tandem array, and how many whole extra
    units it spans (:328-335, :458-462)."""
    if len(repeat) >= K:
        return repeat + repeat[:K - 1], 0
    ext = K - 1 + (len(repeat) - 1) - ((K - 1) % len(repeat))
    s = repeat + (repeat * K)[:ext]
    return s, int(len(s) / len(repeat)) - 1


class _Loop(object):
    """repeatHMM (:313-354): the unit profile closed through the emitting states dummy1 / dummy2."""

    def __init__(self, repeat, pm, probs, tag, std_scale, std_offset):
        P = _merged(dict(skip=.999, leave_repeat=.002), probs)
        unit, self.repeat_offset = tandem_unit(repeat, pm.kmer)
        inner = _Profile(unit, pm, P, tag, True, std_scale, std_offset)
        net = Net(tag + "loop")
        at = net.unite(inner.net)
        d1 = net.node(tag + "dummy1", UNIFORM, (pm.model_min, pm.model_max))
        d2 = net.node(tag + "dummy2", UNIFORM, (pm.model_min, pm.model_max))
        self.s1, self.s2 = at(inner.s1), at(inner.s2)
        net.edge(at(inner.e1), d1, 1); net.edge(at(inner.e2), d2, 1)
        e1 = net.node(tag + "e1", SILENT)           # joins the graph with its first edge (:350)
        net.edge(d1, e1, P["leave_repeat"])
        e2 = net.node(tag + "e2", SILENT)
        net.edge(d2, e2, P["leave_repeat"])
        net.edge(d1, self.s1, 1 - P["leave_repeat"]); net.edge(d2, self.s2, 1 - P["leave_repeat"])
        self.net, self.e1, self.e2, self.d1, self.d2 = net, e1, e2, d1, d2


def flanked_net(repeat, prefix, suffix, pm, config=None):
    """flankedRepeatHMM (:384-431).  Returns (Net, flanking_count, repeat_offset)."""
    P = _merged(dict(skip=1 - 1e-4, seq_std_scale=1.0, rep_std_scale=1.0, seq_std_offset=0.0,
                     rep_std_offset=0.0, e1_ratio=0.1), config if isinstance(config, dict) else None)
    units = int(np.ceil(pm.kmer / len(repeat)))
    pre = _Profile(prefix + (repeat * units)[:-1], pm, P, "prefix", False, P["seq_std_scale"], P["seq_std_offset"])
    suf = _Profile(repeat * units + suffix, pm, P, "suffix", False, P["seq_std_scale"], P["seq_std_offset"])
    rep = _Loop(repeat, pm, P, "repeat", P["rep_std_scale"], P["rep_std_offset"])
    net = Net("flanked")
    a = net.unite(pre.net); r = net.unite(rep.net); z = net.unite(suf.net)      # :417-419
    net.edge(net.start, a(pre.s1), P["e1_ratio"]); net.edge(net.start, a(pre.s2), 1 - P["e1_ratio"])
    net.edge(a(pre.e1), r(rep.s1), 1); net.edge(a(pre.e2), r(rep.s2), 1)
    net.edge(r(rep.e1), z(suf.s1), 1); net.edge(r(rep.e2), z(suf.s2), 1)
    net.edge(z(suf.e1), net.end, 1); net.edge(z(suf.e2), net.end, 1)
    net.counted = (r(rep.d1), r(rep.d2))
    return net, units * 2 - 1, rep.repeat_offset


def mod_net(repeat, pm_base, pm_mod, config=None):
    """repeatModHMM (:447-490).  Returns (Net, model_min, model_max)."""
    P = _merged(dict(rep_std_scale=1.5, rep_std_offset=0.0, leave_repeat=.002), config if isinstance(config, dict) else None)
    unit, _ = tandem_unit(repeat, pm_base.kmer)
    lo = min(pm_base.model_min, pm_mod.model_min); hi = max(pm_base.model_max, pm_mod.model_max)
    base = _Profile(unit, pm_base, P, "base", True, P["rep_std_scale"], P["rep_std_offset"])
    mod = _Profile(unit, pm_mod, P, "mod", True, P["rep_std_scale"] * pm_mod.scale2stdv(pm_base), P["rep_std_offset"])
    net = Net("modification")
    b = net.unite(base.net); m = net.unite(mod.net)
    s0 = net.node("s0", UNIFORM, (lo, hi)); e0 = net.node("e0", UNIFORM, (lo, hi))
    net.edge(net.start, s0, 1)
    for at, prof in ((b, base), (m, mod)):
        net.edge(s0, at(prof.s1), 0.25); net.edge(s0, at(prof.s2), 0.25)
    for at, prof in ((b, base), (m, mod)):
        net.edge(at(prof.e1), e0, 1); net.edge(at(prof.e2), e0, 1)
    net.edge(e0, net.end, P["leave_repeat"]); net.edge(e0, s0, 1 - P["leave_repeat"])
    return net, lo, hi


# ---------------------------------------------------------------------------------------------
# the minimum a decode needs
# ---------------------------------------------------------------------------------------------
class Prepared(object):
    """Arrays strq_oracle_viterbi consumes (viterbi_oracle.c) plus names for path-level rules."""
    pass


def prepare(net, renormalise=True, round_digits=8, normal_form="pomegranate"):
    """Orphan removal, out-edge renormalisation, state order.  No splicing, no edge merging.
    `renormalise`, `round_digits`, `normal_form` exist for tools/residual_probe.py."""
    n = len(net.name)
    alive = [True] * n
    E = [(a, b, math.log(p) if p > 0 else -math.inf) for a, b, p in net.edges()]
    while True:
        has_in, has_out = [False] * n, [False] * n
        for a, b, _ in E:
            has_out[a] = True; has_in[b] = True
        gone = [i for i in range(n) if alive[i] and i not in (net.start, net.end) and not (has_in[i] and has_out[i])]
        if not gone:
            break
        for i in gone:
            alive[i] = False
        E = [e for e in E if alive[e[0]] and alive[e[1]]]
    if renormalise:
        mass = {}
        for a, b, lp in E:
            mass[a] = mass.get(a, 0.0) + math.e ** lp
        for a in mass:
            mass[a] = round(mass[a], round_digits) if round_digits is not None else mass[a]
        E = [(a, b, lp - math.log(mass[a]) if (mass[a] != 1.0 and a != net.end) else lp) for a, b, lp in E]
    emitting = sorted((i for i in range(n) if alive[i] and net.kind[i] != SILENT), key=lambda i: net.name[i])
    silent = [i for i in range(n) if alive[i] and net.kind[i] == SILENT]
    sil = set(silent)
    waits = {i: 0 for i in silent}
    for a, b, _ in E:
        if a in sil and b in sil:
            waits[b] += 1
    order, ready = [], sorted((i for i in silent if waits[i] == 0), key=lambda i: net.name[i])
    while ready:
        i = ready.pop(0); order.append(i)
        for a, b, _ in E:
            if a == i and b in sil:
                waits[b] -= 1
                if waits[b] == 0:
                    ready.append(b)
        ready.sort(key=lambda i: net.name[i])
    if len(order) != len(silent):
        raise ValueError("cycle of silent states")
    final = emitting + order
    pos = {old: k for k, old in enumerate(final)}
    m = len(final)
    ins = [[] for _ in range(m)]
    for a, b, lp in E:                       # in-edges of a state in graph iteration order
        ins[pos[b]].append((pos[a], lp))
    out = Prepared()
    out.n_states, out.silent_start, out.start, out.end = m, len(emitting), pos[net.start], pos[net.end]
    out.in_ptr = np.zeros(m + 1, np.int32)
    src, lps = [], []
    for k in range(m):
        out.in_ptr[k + 1] = out.in_ptr[k] + len(ins[k])
        src += [t[0] for t in ins[k]]; lps += [t[1] for t in ins[k]]
    out.in_src = np.array(src, np.int32); out.in_logp = np.array(lps, np.float64)
    ne = len(emitting)
    out.emis_kind = np.zeros(ne, np.int32)
    out.emis_a, out.emis_b, out.emis_c = np.zeros(ne), np.zeros(ne), np.zeros(ne)
    for k, old in enumerate(emitting):
        out.emis_kind[k] = net.kind[old]
        if net.kind[old] == NORMAL:
            mu, sigma = net.par[old]
            out.emis_a[k] = mu
            out.emis_b[k] = 1.0 / (2 * sigma ** 2)
            out.emis_c[k] = -math.log(sigma * SQRT_2_PI) if normal_form == "pomegranate" else -0.5 * math.log(2 * math.pi * sigma * sigma)
        else:
            lo, hi = net.par[old]
            out.emis_a[k], out.emis_b[k], out.emis_c[k] = lo, hi, -math.log(hi - lo)
    out.count_inc = np.zeros(m, np.int32)
    for i in net.counted:
        out.count_inc[pos[i]] = 1
    out.names = [net.name[i] for i in final]
    out.tag = np.zeros(m, np.int32)
    return out
