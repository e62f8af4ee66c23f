"""utils/buffer.py on the device.

ReplayBuffer_on_policy  :4-50   store / clear / size / compute_advantage (GAE variant G2:
                                float32 recursion, separate dw and done, ddof-1 normalisation) / sample
ReplayBuffer_off_policy :105-135 ring + uniform sampling (np.random.choice(..., replace=False))
Transitions arrive as N-row batches (one vector step) instead of single python tuples; the
on-policy buffer keeps them as a time-major slab [T][N] and flattens env-major on sample so
that every env's trajectory is contiguous, as the single-env reference's is.
"""
import torch

from .. import ops


class ReplayBuffer_on_policy:
    def __init__(self, cfg):
        self.cfg = cfg
        self.buffer = []
        self.samples = None

    def store(self, transitions):
        """transitions = (state[N,D], action[N], reward[N], done[N], dw[N], log_prob[N], value[N], next_value[N])."""
        assert self.samples is None, 'Need to clear the buffer before storing new transitions.'
        self.buffer.append(transitions)

    def clear(self):
        self.buffer = []
        self.samples = None

    def size(self):
        return len(self.buffer) * (self.buffer[0][2].numel() if self.buffer else 0)

    def compute_advantage(self, rewards, dones, dw, values, next_values):
        """:21-35 over [T, N] slabs -> (normalised adv, v_target), both [T, N]."""
        mom = torch.zeros(3, dtype=torch.float64, device=rewards.device)
        adv, v_target = ops.gae_dw(rewards, values, next_values, dones, dw, self.cfg.gamma, self.cfg.lamda,
                                   moments_out=mom)
        adv = ops.normalize_(adv, mom, ddof=1, eps=1e-8)       # torch.std is unbiased (:33)
        return adv, v_target

    def sample(self):
        if self.samples is None:
            cols = list(zip(*self.buffer))
            states = torch.stack(cols[0])                       # [T, N, D]
            actions, rewards, dones, dw, log_probs, values, next_values = (torch.stack(c) for c in cols[1:])
            adv, v_target = self.compute_advantage(rewards.float().contiguous(), dones.to(torch.uint8).contiguous(),
                                                   dw.to(torch.uint8).contiguous(), values.float().contiguous(),
                                                   next_values.float().contiguous())
            T, N = rewards.shape

            def flat(x):
                return x.transpose(0, 1).reshape(T * N, -1)

            self.samples = (flat(states), flat(actions).long(), flat(log_probs), flat(adv), flat(v_target))
        return self.samples


class ReplayBuffer_off_policy:
    def __init__(self, cfg, state_dim=None, action_dim=1, discrete=True):
        self.cfg = cfg
        self.capacity = int(cfg.memory_capacity)
        self.device = torch.device(cfg.device)
        self._ring = None
        self._dims = (state_dim, action_dim, discrete)
        self.cursor, self._size, self.draws = 0, 0, 0

    def _alloc(self, state, action):
        D = state.shape[-1]
        AW = 1 if action.dim() == 1 else action.shape[-1]
        d, c = self.device, self.capacity
        self._ring = (torch.zeros(c, D, device=d), torch.zeros(c, AW, dtype=torch.int32, device=d),
                      torch.zeros(c, device=d), torch.zeros(c, D, device=d), torch.zeros(c, dtype=torch.uint8, device=d))
        self._adtype = torch.int32 if action.dtype in (torch.int32, torch.int64) else torch.float32

    def store(self, transitions):
        """transitions = (state[N,D], action[N(,A)], reward[N], next_state[N,D], done[N])."""
        s, a, r, s2, d = transitions
        if self._ring is None:
            self._alloc(s, a)
        n = r.numel()
        a = a.to(torch.int32) if self._adtype == torch.int32 else a.float().contiguous().view(torch.int32)
        ops.replay_append(self._ring, self.cursor, s.contiguous(), a.reshape(n, -1).contiguous(), r.float().contiguous(),
                          s2.contiguous(), d.to(torch.uint8).contiguous())
        self.cursor = (self.cursor + n) % self.capacity
        self._size = min(self._size + n, self.capacity)

    def clear(self):
        self.cursor = self._size = 0

    def size(self):
        return self._size

    def sample(self):
        """:126-135: a uniform batch; every field float32 (actions too, as the reference casts them)."""
        B = min(int(self.cfg.batch_size), self._size)
        idx = ops.uniform_indices(getattr(self.cfg, "seed", 0) or 0, self.draws, self._size, B, self.device)
        self.draws += 1
        s, a, r, s2, d = ops.replay_gather(self._ring, idx, self._adtype)
        return s, a.float(), r, s2, d


class ReplayBuffer_on_policy_v2:
    """utils/buffer.py:53-102: dense episode-major slabs [batch_size, max_steps, ...] + an `active`
    mask (`dw` defaults to 1), one row per episode, `sample()` trimmed to the longest episode.

    The reference fills one row at a time (`store` ... `next_episode`).  With N env streams every env
    owns the row of its current episode and `next_episode(mask)` hands the finished envs the next free
    rows in env order, so N = 1 reproduces the reference exactly.  The slabs live on `cfg.device`
    (already the `[T][N]`-style device layout `sample()` returns); only tensor indexing is involved,
    no kernels of their own."""

    def __init__(self, cfg, num_envs=1):
        self.cfg, self.N = cfg, int(num_envs)
        self.clear()

    def clear(self):
        E, L, d = int(self.cfg.batch_size), int(self.cfg.max_steps), torch.device(self.cfg.device)
        f = dict(dtype=torch.float32, device=d)
        self.buffer = {
            's': torch.zeros([E, L] + list(self.cfg.state_shape), **f),
            'a': torch.zeros(E, L, dtype=torch.int64, device=d),
            'a_logprob': torch.zeros(E, L, **f), 'r': torch.zeros(E, L, **f), 'd': torch.zeros(E, L, **f),
            'dw': torch.ones(E, L, **f), 'v': torch.zeros(E, L, **f), 'v_': torch.zeros(E, L, **f),
            'active': torch.zeros(E, L, dtype=torch.int8, device=d),
        }
        self.size = torch.zeros(E, dtype=torch.int64, device=d)
        self._row = torch.arange(self.N, dtype=torch.int64, device=d)     # row of env i's current episode
        self._next_free = self.N
        self.episode_num = 0                                             # finished episodes (= current row for N = 1)

    def store(self, transitions):
        """(s, a, r, d, dw, a_logprob, v, v_) with a leading env dimension [N] (scalars accepted for N = 1).
        Envs whose row lies beyond batch_size are dropped (the reference would raise IndexError)."""
        dev = self.size.device
        s, a, r, d, dw, a_logprob, v, v_ = (torch.as_tensor(x, device=dev) for x in transitions)
        row = self._row
        ok = row < self.size.numel()
        row = row[ok]
        pos = self.size[row]
        b = self.buffer
        b['s'][row, pos] = s.reshape(self.N, *b['s'].shape[2:]).float()[ok]
        for key, val in (('a', a), ('a_logprob', a_logprob), ('r', r), ('d', d), ('dw', dw), ('v', v), ('v_', v_)):
            b[key][row, pos] = val.reshape(self.N).to(b[key].dtype)[ok]
        b['active'][row, pos] = 1
        self.size[row] += 1

    def next_episode(self, mask=None):
        """Reference signature for N = 1; `mask` [N] (bool / u8) selects the envs whose episode ended."""
        if mask is None:
            mask = torch.ones(self.N, dtype=torch.bool, device=self.size.device)
        mask = torch.as_tensor(mask, device=self.size.device).bool().reshape(self.N)
        k = int(mask.sum())                                   # host read: rows are a python-side resource
        if k:
            self._row[mask] = self._next_free + torch.arange(k, dtype=torch.int64, device=self.size.device)
            self._next_free += k
            self.episode_num += k

    def sample(self):
        n = int(self.size.max())
        b = self.buffer
        return (b['s'][:, :n], b['a'][:, :n], b['a_logprob'][:, :n], b['r'][:, :n], b['d'][:, :n], b['dw'][:, :n],
                b['v'][:, :n], b['v_'][:, :n], b['active'][:, :n].float())
