"""Small sampler helpers of rlsolver/methods/util.py:498-570 used by the ISCO samplers ([B, N]
elementwise / sort arithmetic on torch tensors; they stay torch ops: control-plane sized, and
they define how torch's generator is consumed)."""
import torch as th


def gumbel(loc):
    uniform_sample = th.rand(loc.shape, device=loc.device)
    return loc - th.log(-th.log(uniform_sample))


def log1mexp(x):
    x = -th.abs(x)
    return th.where(x > -0.693, th.log(-th.expm1(x)), th.log1p(-th.exp(x)))


def noreplacement_sampling_renormalize(ll_idx, dim=-1):
    ll_base = th.max(ll_idx, dim=dim, keepdim=True).values
    prob_idx = th.exp(ll_idx - ll_base)
    ll_delta = th.log(th.cumsum(prob_idx, dim=dim) - prob_idx) + ll_base
    return th.clamp(ll_idx - log1mexp(ll_delta), max=0.0)


def multinomial(log_prob, path_length):
    """Gumbel top-k without replacement, util.py:514-555."""
    num_classes = log_prob.shape[-1]
    perturbed_ll = gumbel(log_prob)
    sorted_ll, _ = th.sort(perturbed_ll)
    threshold = th.gather(sorted_ll, 1, (num_classes - path_length).unsqueeze(1))
    selected_mask = (perturbed_ll >= threshold.expand_as(perturbed_ll)).int()
    selected = {'selected_mask': selected_mask, 'perturbed_ll': perturbed_ll}
    sorted_idx = th.argsort(-perturbed_ll, dim=-1)
    sorted_ll = th.gather(log_prob, dim=-1, index=sorted_idx)
    idx_ll = noreplacement_sampling_renormalize(sorted_ll)
    flat_idx = sorted_idx.view(-1, num_classes)
    flat_ll = idx_ll.view(-1, num_classes)
    ll_selected = th.zeros_like(flat_ll)
    ll_selected.scatter_(1, flat_idx, flat_ll)
    ll_selected = ll_selected.view(log_prob.shape) * selected_mask
    return selected, ll_selected


def bernoulli_logp(log_prob):
    noise = th.rand(log_prob.shape, device=log_prob.device)
    return th.log(noise + 1e-24) < log_prob


def mh_step(log_prob, current_sample, new_sample):
    use_new_sample = bernoulli_logp(log_prob)
    expanded = use_new_sample.unsqueeze(-1).expand_as(new_sample)
    return th.where(expanded, new_sample, current_sample), use_new_sample
