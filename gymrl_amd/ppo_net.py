"""Hand-scheduled forward/backward of the PPO ActorCritic for the update phase.

Default path (hidden_dim 256, `hip_gemm = True`): every contraction is a hand-written exact-f32 MFMA kernel of
csrc/gemm.hip and the loss sits inside the heads pass — `step()`:

    forward   H1  = tanh(X W1^T + b1)                 gymrl_linear_tanh_smallk   (K = obs_dim)
              H2  = tanh(H1 W2^T + b2)                gymrl_linear_fwd, tanh in the MFMA epilogue
              Zac = H2 [Wa1;Wc1]^T + [ba1;bc1]        gymrl_linear_fwd (N = 512, bias only: the heads pass applies
                                                      the tanh where the VALU is idle)
    loss+heads  dZac (in place), dbac, dWa2, dba2, dWc2, dbc2, metrics
                                                      gymrl_heads_loss_fwd_bwd: ONE pass over Zac
    backward  d[Wa1;Wc1] = dZac^T H2                  gymrl_linear_bwd_weight (N = 512)
              dZ2 = (dZac [Wa1;Wc1]) (1 - H2^2)       gymrl_linear_bwd_input, tanh' in the epilogue
              dW2 = dZ2^T H1, db2 = colsum dZ2        gymrl_linear_bwd_weight (+ bias gradient from the same pass)
              dZ1 = (dZ2 W2) (1 - H1^2)               gymrl_linear_bwd_input
              dW1 = dZ1^T X, db1 = colsum dZ1         gymrl_linear_smallk_bwd (dZ given)

The round-1 path below (library GEMMs + separate HBM passes, `forward()` / `backward()`) stays for other widths and
as the A/B baseline of tools/micro_update.py.

`ActorCritic.evaluate_actions` + `loss.backward()` (ppo_lunarlander.py:110-117, :303) on a
B = 262,144-row minibatch is, under autograd, 10 library GEMMs plus ~30 elementwise /
reduction launches that each stream a [B, 256] activation through HBM.  Here the same math
is scheduled by hand:

    forward   H1  = tanh(X W1^T + b1)                 gymrl_linear_tanh_smallk   (K = obs_dim)
              H2  = tanh(H1 W2^T + b2)                addmm (hipBLASLt) + gymrl_tanh_inplace
              Zac = H2 [Wa1;Wc1]^T + [ba1;bc1]        ONE GEMM for actor.0 and critic.0 (N = 512)
              logits = tanh(Za) Wa2^T + ba2, v = tanh(Zc) Wc2^T + bc2
                                                      gymrl_heads_fwd_tanh: one READ-ONLY pass over Zac, heads from
                                                      registers; tanh(Zac) is not written back
    backward  dZac, dbac, dWa2, dba2, dWc2, dbc2      gymrl_heads_bwd: one pass over Zac (recomputes the same tanh,
                                                      writes dZac in place over the pre-activations)
              d[Wa1;Wc1] = dZac^T H2                  split-K batched GEMM
              dH2 = dZac [Wa1;Wc1]                    ONE GEMM (K = 512): no gradient accumulation
              dZ2 = dH2 (1 - H2^2), db2               gymrl_tanh_bwd_colsum (in place)
              dW2 = dZ2^T H1, dH1 = dZ2 W2            GEMMs
              dW1, db1                                gymrl_linear_smallk_bwd (dZ1 never stored)

The parameters stay the module's own nn.Parameters (state_dict-compatible with the reference);
`flatten_module(order=LAYOUT)` only places actor.0 / critic.0 next to each other in the flat
buffer so that [Wa1;Wc1] and [ba1;bc1] are plain views.  Gradients are written (not
accumulated) into the parameters' .grad views of the flat gradient buffer.
"""
import torch

from . import ops

# flat-buffer order that makes [actor.0 ; critic.0] contiguous (weights, then biases)
LAYOUT = ["shared.0.weight", "shared.0.bias", "shared.2.weight", "shared.2.bias",
          "actor.0.weight", "critic.0.weight", "actor.0.bias", "critic.0.bias",
          "actor.2.weight", "actor.2.bias", "critic.2.weight", "critic.2.bias"]


def supported(model):
    H = model.shared[0].out_features
    D = model.shared[0].in_features
    A = model.actor[2].out_features
    # multiples of 64 keep flatten_module's 64-float padding from separating actor.0 / critic.0
    return (H in (64, 128, 256) and D in (2, 3, 4, 8) and A in (2, 4)
            and model.shared[2].out_features == H and model.actor[0].out_features == H
            and model.critic[0].out_features == H)


class FusedActorCriticUpdate:
    def __init__(self, model, max_rows):
        if not supported(model):
            raise ValueError("unsupported ActorCritic shape for the fused update path")
        self.m = model
        W1 = model.shared[0].weight
        dev = W1.device
        self.H, self.D, self.A = W1.shape[0], W1.shape[1], model.actor[2].out_features
        H = self.H
        wa, wc = model.actor[0].weight, model.critic[0].weight
        ba, bc = model.actor[0].bias, model.critic[0].bias
        if wc.data_ptr() != wa.data_ptr() + wa.numel() * 4 or bc.data_ptr() != ba.data_ptr() + ba.numel() * 4:
            raise ValueError("actor.0 / critic.0 are not adjacent: flatten_module(model, device, order=ppo_net.LAYOUT)")
        flat, gflat = model._flat_params, model._flat_grads

        def view(p, n, shape, buf):
            off = (p.data_ptr() - flat.data_ptr()) // 4
            return buf[off:off + n].view(shape)
        self.Wac = view(wa, 2 * H * H, (2 * H, H), flat)
        self.bac = view(ba, 2 * H, (2 * H,), flat)
        self.dWac = view(wa, 2 * H * H, (2 * H, H), gflat)
        self.dbac = view(ba, 2 * H, (2 * H,), gflat)
        self.ws = ops.mlp_train_workspace(H, self.D, self.A, dev)
        R = int(max_rows)
        self.R = R
        self.H1 = torch.empty(R, H, device=dev)
        self.H2 = torch.empty(R, H, device=dev)
        self.Hac = torch.empty(R, 2 * H, device=dev)
        self._dZac = None            # only allocated when heads_bwd does not run in place
        self.dH2 = torch.empty(R, H, device=dev)
        self.dH1 = torch.empty(R, H, device=dev)
        self.logits = torch.empty(R, self.A, device=dev)
        self.value = torch.empty(R, 1, device=dev)
        self._x = None
        self._pre, self._pre_bias = False, None
        self.timers = None      # bench.py: KernelTimers bracketing the hand-written HBM passes
        self.fused_heads_forward = True
        self.recompute_tanh = True      # heads_fwd does not store tanh(Zac); heads_bwd recomputes it (2 KB/row less traffic)
        self.bwd_in_place = True        # heads_bwd writes dZac over the pre-activations it just read (they are dead after it)
        self.recompute_h1 = False       # linear_smallk_bwd recomputing H1 from x: 1 KB/row less traffic but 102 -> 115 us
                                        # (the pass is issue-bound once H1 is not read), measured and left off
        self.bias_in_gemm = False
        self.overlap_dw = False     # dW GEMMs on a side stream under the HBM passes: measured 3.255 vs 3.23 ms, no gain
        self._side = torch.cuda.Stream(device=dev)
        # hand-written MFMA GEMMs + loss inside the heads pass (module docstring); needs hidden_dim == 256
        self.hip_gemm = H == 256 and self.A in (2, 4)
        self.gemm_ws = ops.gemm_workspace(dev) if self.hip_gemm else None

    def _timed(self, name, units, fn, *args):
        tm = self.timers
        if tm is None:
            return fn(*args)
        tm.start(name)
        r = fn(*args)
        tm.stop(name, units)
        return r

    @torch.no_grad()
    def forward(self, x):
        """x [B, obs] -> (logits [B, A], values [B]); keeps the activations backward() needs."""
        m, B, H = self.m, x.shape[0], self.H
        if B > self.R:
            raise ValueError("minibatch larger than the buffers")
        H1, H2, Hac = self.H1[:B], self.H2[:B], self.Hac[:B]
        self._timed("linear_tanh_smallk", B, ops.linear_tanh_smallk, x, m.shared[0].weight, m.shared[0].bias, H1)
        if self.bias_in_gemm:                     # library GEMM with its bias epilogue
            torch.addmm(m.shared[2].bias, H1, m.shared[2].weight.t(), out=H2)
            self._timed("tanh_inplace", H2.numel(), ops.tanh_inplace, H2)
            torch.addmm(self.bac, H2, self.Wac.t(), out=Hac)
            bac = None
        else:                                     # plain GEMMs; the biases ride on the passes that follow anyway
            torch.mm(H1, m.shared[2].weight.t(), out=H2)
            self._timed("tanh_inplace", H2.numel(), ops.tanh_inplace, H2, m.shared[2].bias)
            torch.mm(H2, self.Wac.t(), out=Hac)
            bac = self.bac
        logits, value = self.logits[:B], self.value[:B]
        if self.fused_heads_forward:
            self._pre = self.recompute_tanh
            self._timed("heads_fwd_tanh", B, ops.heads_fwd_tanh, Hac, m.actor[2].weight, m.actor[2].bias,
                        m.critic[2].weight, m.critic[2].bias, logits, value, bac, not self._pre)
            self._pre_bias = bac
        else:                                     # tanh pass + two skinny library GEMMs on views of Hac
            self._pre = False
            self._timed("tanh_inplace", Hac.numel(), ops.tanh_inplace, Hac, bac)
            torch.addmm(m.actor[2].bias, Hac[:, :H], m.actor[2].weight.t(), out=logits)
            torch.addmm(m.critic[2].bias, Hac[:, H:], m.critic[2].weight.t(), out=value)
        self._x = x
        return logits, value.view(-1)

    def metric_blocks(self, B):
        """Rows of the f64[blocks, 5] metric-partials buffer step() fills for a minibatch of B rows."""
        return ops.heads_loss_blocks(B, self.H)

    @torch.no_grad()
    def step(self, x, act, logp_old, adv, ret, loss_cfg, adv_moments, metric_parts, reducer=None):
        """Forward, loss and backward of one minibatch (module docstring): writes every parameter gradient
        (overwrite) and the f64[metric_blocks(B), 5] partial metric sums.  `reducer` (dist.GradReducer with its
        split at actor.0.weight): bucket 1 = [actor.0 | critic.0, heads] is launched as soon as its last writer is
        queued, bucket 0 = the trunk after the last kernel."""
        m, B = self.m, x.shape[0]
        if B > self.R:
            raise ValueError("minibatch larger than the buffers")
        t = self._timed
        H1, H2, Zac, dZ2, dZ1 = self.H1[:B], self.H2[:B], self.Hac[:B], self.dH2[:B], self.dH1[:B]
        W2, ws = m.shared[2].weight, self.gemm_ws
        t("linear_tanh_smallk", B, ops.linear_tanh_smallk, x, m.shared[0].weight, m.shared[0].bias, H1)
        t("gemm_fwd_256_tanh", B, ops.linear_fwd, H1, W2, m.shared[2].bias, H2, True)
        t("gemm_fwd_512", B, ops.linear_fwd, H2, self.Wac, self.bac, Zac, False)
        t("heads_loss_fwd_bwd", B, ops.heads_loss_fwd_bwd, Zac, None, m.actor[2].weight, m.actor[2].bias,
          m.critic[2].weight, m.critic[2].bias, act, logp_old, adv, ret, loss_cfg, adv_moments, self.dbac,
          m.actor[2].weight.grad, m.actor[2].bias.grad, m.critic[2].weight.grad, m.critic[2].bias.grad, metric_parts, self.ws)
        t("gemm_dw_512", B, ops.linear_bwd_weight, Zac, H2, self.dWac, ws)
        if reducer is not None:
            reducer.launch(1)
        t("gemm_dx_512_tanhbwd", B, ops.linear_bwd_input, Zac, self.Wac, H2, dZ2)
        t("gemm_dw_256_db", B, ops.linear_bwd_weight, dZ2, H1, W2.grad, ws, m.shared[2].bias.grad)
        t("gemm_dx_256_tanhbwd", B, ops.linear_bwd_input, dZ2, W2, H1, dZ1)
        t("linear_smallk_bwd", B, ops.linear_smallk_bwd, dZ1, None, x, m.shared[0].weight.grad, m.shared[0].bias.grad,
          self.ws)
        if reducer is not None:
            reducer.launch(0)

    def _recompute_h1(self):
        """(W1, b1) when linear_smallk_bwd recomputes H1 from the observations instead of reading it."""
        m = self.m
        return (m.shared[0].weight, m.shared[0].bias) if self.recompute_h1 else (None, None)

    @staticmethod
    def _dw(dy, x, out):
        """out = dy^T x with the reduction dimension (rows) split into independent slices."""
        B, N = dy.shape
        K = x.shape[1]
        S = 128
        while S > 1 and (B % S or B // S < 64):
            S //= 2
        if S == 1:
            return torch.mm(dy.t(), x, out=out)
        tmp = torch.bmm(dy.view(S, B // S, N).transpose(1, 2), x.view(S, B // S, K))
        return torch.sum(tmp, 0, out=out)

    @torch.no_grad()
    def backward(self, dlogits, dvalues):
        """Writes every parameter gradient (overwrite, not accumulate) from dL/dlogits, dL/dvalues."""
        m, x, H = self.m, self._x, self.H
        B = x.shape[0]
        H1, H2, Hac = self.H1[:B], self.H2[:B], self.Hac[:B]
        if not self.bwd_in_place and self._dZac is None:
            self._dZac = torch.empty_like(self.Hac)
        dZac, dH2, dH1 = (Hac if self.bwd_in_place else self._dZac[:B]), self.dH2[:B], self.dH1[:B]
        self._timed("heads_bwd", B, ops.heads_bwd, Hac, dlogits, dvalues.view(-1), m.actor[2].weight,
                    m.critic[2].weight, dZac, self.dbac, m.actor[2].weight.grad, m.actor[2].bias.grad,
                    m.critic[2].weight.grad, m.critic[2].bias.grad, self.ws, self._pre,
                    self._pre_bias if self._pre else None)
        if not self.overlap_dw:
            self._dw(dZac, H2, self.dWac)
            torch.mm(dZac, self.Wac, out=dH2)
            self._timed("tanh_bwd_colsum", B, ops.tanh_bwd_colsum, dH2, H2, m.shared[2].bias.grad, self.ws)
            self._dw(dH2, H1, m.shared[2].weight.grad)
            torch.mm(dH2, m.shared[2].weight, out=dH1)
            self._timed("linear_smallk_bwd", B, ops.linear_smallk_bwd, dH1, H1, x, m.shared[0].weight.grad,
                        m.shared[0].bias.grad, self.ws, *self._recompute_h1())
            return
        # The weight-gradient GEMMs (MFMA-bound, off the critical path) run on a side stream under the
        # HBM-bound passes of the main stream: dWac under tanh_bwd_colsum, dW2 under linear_smallk_bwd.
        main, side = torch.cuda.current_stream(), self._side
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            self._dw(dZac, H2, self.dWac)
        torch.mm(dZac, self.Wac, out=dH2)
        self._timed("tanh_bwd_colsum", B, ops.tanh_bwd_colsum, dH2, H2, m.shared[2].bias.grad, self.ws)
        ev2 = torch.cuda.Event()
        ev2.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev2)
            self._dw(dH2, H1, m.shared[2].weight.grad)
        torch.mm(dH2, m.shared[2].weight, out=dH1)
        self._timed("linear_smallk_bwd", B, ops.linear_smallk_bwd, dH1, H1, x, m.shared[0].weight.grad,
                    m.shared[0].bias.grad, self.ws, *self._recompute_h1())
        main.wait_stream(side)                     # the optimiser and the next forward see every gradient / free buffer
