"""CPU ORACLE for the ISCO sampler steps -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in numpy float32 and in the reference's own shape (full-row sorts, cumulative sums over all N
entries, scatter back), the op chains of
    rlsolver/envs/env_ISCO.py:26-77     ISCO_maxcut.step / proposal / get_local_dist / ll_y2x
    rlsolver/envs/env_ISCO.py:188-236   ISCO_TSP.step / proposal / get_local_dist / y2x
    rlsolver/envs/env_ISCO.py:238-344   ISCO_TSP.opt_2 / switch
    rlsolver/methods/util.py:498-570    gumbel, log1mexp, noreplacement_sampling_renormalize, multinomial,
                                        bernoulli_logp, mh_step
with every random draw passed in.  Pinned against tests/golden/isco_steps.npz (captured from the reference with
its torch draws recorded): see tests/test_oracle_isco.py.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=F))).astype(F)


def log1mexp(x):
    """util.py:502-505"""
    x = -np.abs(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(x > F(-0.693), np.log(-np.expm1(x)), np.log1p(-np.exp(x))).astype(F)


def noreplacement_sampling_renormalize(ll_idx):
    """util.py:507-512"""
    ll_base = ll_idx.max(axis=-1, keepdims=True)
    prob = np.exp(ll_idx - ll_base).astype(F)
    with np.errstate(divide="ignore"):
        ll_delta = (np.log(np.cumsum(prob, axis=-1, dtype=F) - prob) + ll_base).astype(F)
    return np.minimum(ll_idx - log1mexp(ll_delta), F(0)).astype(F)


def multinomial(log_prob, path_length, u):
    """util.py:514-555 -> (selected_mask int [B, N], perturbed_ll [B, N], ll_selected [B, N])"""
    B, N = log_prob.shape
    perturbed = (log_prob - np.log(-np.log(u))).astype(F)                     # gumbel, util.py:498-500
    sorted_ll = np.sort(perturbed, axis=-1)
    threshold = sorted_ll[np.arange(B), N - np.asarray(path_length)][:, None]
    mask = (perturbed >= threshold).astype(np.int32)
    sorted_idx = np.argsort(-perturbed, axis=-1, kind="stable")
    idx_ll = noreplacement_sampling_renormalize(np.take_along_axis(log_prob, sorted_idx, axis=-1))
    ll_selected = np.zeros_like(idx_ll)
    np.put_along_axis(ll_selected, sorted_idx, idx_ll, axis=-1)
    return mask, perturbed, (ll_selected * mask).astype(F)


def reverse_ll(log_prob, mask, perturbed):
    """ll_y2x / y2x, env_ISCO.py:65-77, :214-226"""
    backwd_idx = np.argsort(perturbed, axis=-1, kind="stable")
    lp = np.where(mask.astype(bool), log_prob, F(-1e18)).astype(F)
    backwd_ll = np.take_along_axis(lp, backwd_idx, axis=-1)
    backwd_mask = np.take_along_axis(mask, backwd_idx, axis=-1)
    ll_backwd = noreplacement_sampling_renormalize(backwd_ll)
    return np.where(backwd_mask.astype(bool), ll_backwd, F(0)).sum(axis=-1, dtype=F)


def remaining_mass(log_prob, mask, perturbed, forward=True):
    """Conditioning of the renormalisation terms (test gating only, float64): the smallest probability mass still
    undrawn before any of the selected draws, in forward (descending perturbed value, normalised over all entries)
    or reverse (ascending, normalised over the selected entries' maximum as the reference does) order."""
    B = log_prob.shape[0]
    out = np.ones(B)
    for b in range(B):
        sel = np.flatnonzero(mask[b])
        order = sel[np.argsort(-perturbed[b, sel] if forward else perturbed[b, sel], kind="stable")]
        lp = log_prob[b, order].astype(np.float64)
        base = log_prob[b].max() if forward else lp.max()
        p = np.exp(lp - base)
        before = np.concatenate([[0.0], np.cumsum(p)[:-1]])
        # the term is log(1 - exp(log(before) + base)) = log(1 - before * exp(base))
        out[b] = np.min(1.0 - before * np.exp(float(base)))
    return out


def mh_accept(log_acc, u):
    """bernoulli_logp, util.py:556-559"""
    with np.errstate(divide="ignore"):
        return np.log((u + F(1e-24)).astype(F)) < log_acc


# ------------------------------------------------------------------------------------------------- MaxCut
def maxcut_local_dist(x01, eu, ev, temperature):
    """get_local_dist, env_ISCO.py:51-63 with the energy of :79-86: energy = #cut / T, score_i = (1 - 2 x_i) grad_i / 2
    where grad_i = dE/dx_i = -sum_{j ~ i} (2 x_j - 1) / T  (what autograd returns)."""
    T = F(temperature)
    d = (x01 * 2 - 1).astype(F)
    is_cut = (1 - d[:, eu] * d[:, ev]) / F(2)
    energy = (is_cut.sum(axis=-1, dtype=F) / T).astype(F)
    grad = np.zeros_like(d)
    for b in range(x01.shape[0]):
        np.add.at(grad[b], eu, -d[b, ev] / T)
        np.add.at(grad[b], ev, -d[b, eu] / T)
    score = ((1 - x01 * 2) * grad / F(2)).astype(F)
    return energy, log_softmax(score)


def maxcut_step(x01, eu, ev, path_length, temperature, u_gumbel, u_accept):
    """ISCO_maxcut.step, env_ISCO.py:26-35.  Returns a dict of every intermediate the golden fixture holds."""
    x01 = x01.astype(F)
    ll_x, log_prob = maxcut_local_dist(x01, eu, ev, temperature)
    mask, perturbed, ll_sel = multinomial(log_prob, path_length, u_gumbel)
    y = (x01 * (1 - mask) + mask * (1 - x01)).astype(F)
    ll_x2y = ll_sel.sum(axis=-1, dtype=F)
    ll_y, log_prob_y = maxcut_local_dist(y, eu, ev, temperature)
    ll_y2x = reverse_ll(log_prob_y, mask, perturbed)
    log_acc = np.minimum(ll_y + ll_y2x - ll_x - ll_x2y, F(0)).astype(F)
    use = mh_accept(log_acc, u_accept)
    out = np.where(use[:, None], y, x01)
    cond = np.minimum(remaining_mass(log_prob, mask, perturbed, True), remaining_mass(log_prob_y, mask, perturbed, False))
    return dict(ll_x=ll_x, ll_x2y=ll_x2y, mask=mask, y_prop=y, ll_y=ll_y, ll_y2x=ll_y2x, log_acc=log_acc, y=out,
                energy=(ll_y * F(temperature)).astype(F), acc=np.exp(log_acc).astype(F), remaining_mass=cond,
                accept_margin=np.abs(np.log((u_accept + F(1e-24)).astype(np.float64)) - log_acc))


# ------------------------------------------------------------------------------------------------- TSP
def tsp_opt_2(sample, dist, nearest, random, K, temperature, u_partner, r_near, r_rand):
    """opt_2, env_ISCO.py:238-335 -> (-delta / T, indices, ban)"""
    B, N = sample.shape
    cond = u_partner < (K / (K + 1))
    near = nearest[sample, r_near]
    far = random[sample, r_rand]
    sel = np.where(cond, near, far)
    order = np.argsort(sample, axis=1, kind="stable")                        # sort + searchsorted == inverse permutation
    indices = np.take_along_axis(order, sel, axis=1)
    pos = np.arange(N)[None, :].repeat(B, 0)
    g = lambda idx: np.take_along_axis(sample, idx % N, axis=1)
    s_m1, s_m0 = g(pos + 1), g(pos - 1)
    ban = (s_m1 == sel) | (s_m0 == sel)
    s_i0, s_i1, s_i = g(indices - 1), g(indices + 1), g(indices)
    c3 = s_m1 == s_i0
    nm, nm1, nm2 = g(pos), s_m1, g(pos + 2)
    D = lambda u, v: dist[u, v]
    d3 = -(D(nm, nm1) + D(s_i, s_i1)) + (D(nm, s_i) + D(s_i0, s_i1))
    d4 = -(D(nm, nm1) + D(nm1, nm2) + D(s_i0, s_i) + D(s_i, s_i1)) + (D(nm, s_i) + D(s_i, nm2) + D(s_i0, nm1) + D(nm1, s_i1))
    delta = np.where(ban, F(0), np.where(c3, d3, d4)).astype(F)
    return (-delta / F(temperature)).astype(F), indices, ban


def tsp_step(x, dist, nearest, random, K, path_length, temperature, u_partner, r_near, r_rand, u_gumbel, u_accept):
    """ISCO_TSP.step, env_ISCO.py:188-201 -> dict(log_acc, cur_x, y, mean_acc)"""
    B, N = x.shape
    cur = x.copy()
    traj = np.zeros((B, 3, path_length), F)
    one = np.ones(B, np.int64)
    for i in range(path_length):
        logratio, indices, ban = tsp_opt_2(cur, dist, nearest, random, K, temperature, u_partner[i], r_near[i], r_rand[i])
        logratio = np.where(ban, F(-1e6), logratio).astype(F)
        logits = (logratio / F(2)).astype(F)
        log_prob = log_softmax(logits)
        mask, perturbed, ll_sel = multinomial(log_prob, one, u_gumbel[i])
        logits2 = (logits * (1 - 2 * mask)).astype(F)
        env, posn = np.nonzero((mask == 1) & ~ban)
        nxt = cur.copy()
        j = indices[env, posn]
        p1 = (posn + 1) % N
        tmp = nxt[env, p1].copy()
        nxt[env, p1] = nxt[env, j]
        nxt[env, j] = tmp
        traj[:, 0, i] = (logratio * mask).sum(axis=-1, dtype=F)
        traj[:, 1, i] = -ll_sel.sum(axis=-1, dtype=F)
        traj[:, 2, i] = reverse_ll(log_softmax(logits2), mask, perturbed)
        cur = nxt
    log_acc = np.minimum(traj.sum(axis=(1, 2), dtype=F), F(0)).astype(F)
    use = mh_accept(log_acc, u_accept)
    y = np.where(use[:, None], cur, x)
    return dict(log_acc=log_acc, cur_x=cur, y=y, mean_acc=np.exp(log_acc).astype(F).mean(dtype=F))
