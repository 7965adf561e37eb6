"""Seeded logits rows for the sampler fixtures (tests/golden/sampler.npz): the generator (make_golden.py::gen_sampler, which runs
the REFERENCE's ``top_p_sampling`` on them) and the tests build the same rows from (seed, shape, recipe), so only the expected
distributions are stored.  numpy's legacy RandomState stream is frozen across numpy versions."""
import numpy as np

# name: (seed, N, recipe, dtype, top_k, top_p, temperature)
CASES = {
    "vocab_default":   (9001, 65024, "normal3", "f16", 100, 0.8, 1.0),     # the reference's defaults on ChatGLM2's vocabulary
    "vocab_f32_t07":   (9002, 65024, "normal3", "f32", 50, 0.95, 0.7),
    "vocab_peaked":    (9003, 65024, "peaked", "f16", 100, 0.8, 1.0),      # a few tokens hold the mass: the top-p cut bites early
    "vocab_flat_k256": (9004, 65024, "flat", "f16", 256, 0.9, 1.3),        # near-uniform: all top_k entries survive
    "vocab_bf16":      (9005, 65024, "normal3", "bf16", 100, 0.8, 1.0),    # bf16 logits: 8 mantissa bits, thousands of exact ties
    "small_n":         (9006, 1000, "normal3", "f32", 100, 0.8, 1.0),
    "tiny_n_lt_k":     (9007, 37, "normal3", "f16", 100, 0.8, 1.0),        # fewer logits than top_k
    "ties_coarse":     (9008, 65024, "coarse", "f16", 100, 0.8, 1.0),      # logits on a 0.5 grid: ties across the k-th place
    "masked_inf":      (9009, 65024, "masked", "f16", 100, 0.8, 1.0),      # all but 40 logits at -inf (constrained decoding)
    "masked_500":      (9015, 65024, "masked500", "f16", 100, 0.8, 1.0),   # 500 logits allowed: more than top_k finite values among the -inf
    "sorted_ascending": (9010, 65024, "ascending", "f32", 100, 0.8, 1.0),  # defeats the first threshold: the bisection path
    "all_equal":       (9011, 4096, "equal", "f16", 100, 0.8, 1.0),        # one tie group
    "top_k_1":         (9012, 65024, "normal3", "f16", 1, 0.8, 1.0),
    "k1024":           (9013, 65024, "normal3", "f16", 1024, 0.99, 2.0),
    "ragged_n":        (9014, 65021, "normal3", "f16", 100, 0.8, 1.0),     # N % 8 != 0
}


def logits_for(name: str) -> np.ndarray:
    """float32 array holding values exactly representable in the case's dtype."""
    seed, N, recipe, dtype, *_ = CASES[name]
    rs = np.random.RandomState(seed)
    if recipe == "normal3":
        x = rs.standard_normal(N) * 3.0
    elif recipe == "peaked":
        x = rs.standard_normal(N) * 1.5
        x[rs.randint(0, N, 5)] += 14.0
    elif recipe == "flat":
        x = rs.standard_normal(N) * 0.05
    elif recipe == "coarse":
        x = np.round(rs.standard_normal(N) * 3.0 * 2.0) / 2.0
    elif recipe == "masked":
        x = np.full(N, -np.inf)
        keep = rs.choice(N, 40, replace=False)
        x[keep] = rs.standard_normal(40) * 2.0
    elif recipe == "masked500":
        x = np.full(N, -np.inf)
        keep = rs.choice(N, 500, replace=False)
        x[keep] = rs.standard_normal(500) * 2.0
    elif recipe == "ascending":
        x = np.sort(rs.standard_normal(N) * 3.0)
    elif recipe == "equal":
        x = np.full(N, 1.25)
    else:
        raise KeyError(recipe)
    x = x.astype(np.float32)
    if dtype == "f16":
        x = x.astype(np.float16).astype(np.float32)
    elif dtype == "bf16":
        u = x.view(np.uint32)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        x = u.astype(np.uint32).view(np.float32)
    return x
