"""TEST INFRASTRUCTURE (oracle): CPU restatement of the counter-based random stream of the stochastic layers.

The reference's nn.Dropout / DropPath (/root/reference/models/changeformer.py:107,130-132,160-162,203-207,236-241; timm
drop_path) draw Bernoulli masks from torch's global generator.  The HIP path (kurosiwo_amd/csrc/common.h ksmi_rng_*,
stochastic.hip, attn_mfma.hip, cformer.hip) replaces that generator by a pure function of (seed, step, site, element index);
this file restates that function in numpy so that the oracle -- and the reference's own modules, see
oracle/gen_golden.py:gen_changeformer_drop -- can be run with exactly the masks the kernels regenerate.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
# sites of one encoder block (site id = 8 * global block index + one of these): kurosiwo_amd/changeformer_plan.py
SITE_ATTN, SITE_PROJ, SITE_MLP1, SITE_MLP2, SITE_PATH_ATTN, SITE_PATH_MLP = range(6)


def mix32(x):
    """ksmi_mix32: 32-bit avalanche hash (xor-shift / multiply rounds), vectorised"""
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x21F0AAAD)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x735A2D97)) & M32
    x ^= x >> np.uint64(15)
    return x


def rng_key(seed, step, site):
    """ksmi_rng_key"""
    a = mix32((int(site) + 0x9E3779B9) & 0xFFFFFFFF)
    b = mix32(np.uint64(int(step) & 0xFFFFFFFF) ^ a)
    return mix32(np.uint64(int(seed) & 0xFFFFFFFF) ^ b)


def draws(seed, step, site, first, count):
    """ksmi_rng_u32 for the element indices first .. first+count-1"""
    idx = (np.arange(count, dtype=np.uint64) + np.uint64(first)) & M32
    return mix32(mix32(idx) ^ rng_key(seed, step, site))


def threshold(p):
    """(thr, inv_keep) exactly as kurosiwo_amd/changeformer_plan.py:drop_threshold"""
    if p <= 0.0:
        return 0, 1.0
    return min(0xFFFFFFFF, int(round(p * 4294967296.0))), 1.0 / (1.0 - p)


def scale_mask(seed, step, site, p, first, shape):
    """float32 array of `shape`: inv_keep where the element (row-major index first + i) is kept, else 0"""
    thr, inv = threshold(p)
    n = int(np.prod(shape))
    if thr == 0:
        return np.ones(shape, dtype=np.float32)
    keep = draws(seed, step, site, first, n) >= np.uint64(thr)
    return (keep.astype(np.float32) * np.float32(inv)).reshape(shape)


class DropStream:
    """The stochastic layers of one training forward: probabilities (ChangeFormerV6.__init__ :651-653) and the stream position."""

    def __init__(self, seed, step, drop_rate=0.1, attn_drop=0.1, drop_path_rate=0.1, nblocks=13):
        self.seed, self.step = seed, step
        self.p_drop, self.p_attn = drop_rate, attn_drop
        self.dpr = [drop_path_rate * i / (nblocks - 1) for i in range(nblocks)]     # torch.linspace(0, drop_path_rate, sum(depths))

    def elements(self, gi, site, p, first, shape):
        return scale_mask(self.seed, self.step, 8 * gi + site, p, first, shape)

    def path(self, gi, site, sample0, nsamples):
        """DropPath: one draw per sample (index sample0 + b in the 2B-image batch of the HIP path)"""
        return scale_mask(self.seed, self.step, 8 * gi + site, self.dpr[gi], sample0, (nsamples,))
