"""Hand-scheduled forward / loss / backward of the PPO ActorCritic for the update phase — ONE path:
`FusedActorCriticUpdate.step()`.  Every contraction is a hand-written exact-f32 MFMA kernel of csrc/gemm.hip (hidden
widths 64 / 128 / 256) and every elementwise / reduction pass an HBM-bound kernel of csrc/mlp_train.hip; autograd and
the GEMM libraries are not on this path.

    forward   H1  = tanh(X W1^T + b1)                 gymrl_linear_tanh_smallk   (K = obs_dim)
              H2  = tanh(H1 W2^T + b2)                gymrl_linear_fwd, tanh in the MFMA epilogue
              Zac = H2 [Wa1;Wc1]^T + [ba1;bc1]        gymrl_linear_fwd (N = 2 hidden, bias only: the heads pass applies
                                                      the tanh where the VALU is idle)
    loss+heads  dZac (in place), dbac, dWa2, dba2, dWc2, dbc2, metrics
                                                      hidden 256: gymrl_heads_loss_fwd_bwd, ONE pass over Zac;
                                                      hidden 64 / 128: gymrl_heads_fwd_tanh -> gymrl_ppo_loss_fwd_bwd ->
                                                      gymrl_heads_bwd (the same per-row arithmetic as three passes)
    backward  d[Wa1;Wc1] = dZac^T H2                  gymrl_linear_bwd_weight (hidden 256) / gymrl_lin_bwd_weight
              dZ2 = (dZac [Wa1;Wc1]) (1 - H2^2)       gymrl_linear_bwd_input, tanh' in the epilogue
              dW2 = dZ2^T H1, db2 = colsum dZ2        gymrl_linear_bwd_weight / gymrl_lin_bwd_weight (+ bias gradient)
              dZ1 = (dZ2 W2) (1 - H1^2)               gymrl_linear_bwd_input
              dW1 = dZ1^T X, db1 = colsum dZ1         gymrl_linear_smallk_bwd (dZ given)

`ActorCritic.evaluate_actions` + `loss.backward()` (ppo_lunarlander.py:110-117, :303) on a B = 262,144-row minibatch is,
under autograd, 10 library GEMMs plus ~30 elementwise / reduction launches that each stream a [B, 256] activation
through HBM.  (The round-1 schedule — library GEMMs + separate passes — lives in tools/legacy_update_path.py as the A/B
baseline of tools/micro_update.py.)

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
        self.dH2 = torch.empty(R, H, device=dev)
        self.dH1 = torch.empty(R, H, device=dev)
        self.timers = None      # bench.py: KernelTimers bracketing every launch
        self.one_pass_heads = H == 256           # gymrl_heads_loss_fwd_bwd (a row = one 64-lane load of 4 columns per lane)
        self.gemm_ws = ops.gemm_workspace(dev) if H == 256 else None
        # the five batch reductions' second halves as ONE launch at the end of step() (gymrl_update_finalize) instead of five
        # behind their producers: every producer then keeps its partials in a workspace of its own until that launch
        self.defer_finalize = H == 256
        self.gemm_ws2 = ops.gemm_workspace(dev) if H == 256 else None
        self.ws2 = ops.mlp_train_workspace(H, self.D, self.A, dev) if H == 256 else None
        if not self.one_pass_heads:
            self.logits = torch.empty(R, self.A, device=dev)
            self.value = torch.empty(R, 1, device=dev)
            self.dlogits = torch.empty(R, self.A, device=dev)
            self.dvalue = torch.empty(R, device=dev)
            self.lin_ws = ops.lin_workspace(R, 2 * H, H, 1, dev) if R > 512 else None

    def _timed(self, name, units, fn, *args, **kw):
        tm = self.timers
        if tm is None:
            return fn(*args, **kw)
        tm.start(name)
        r = fn(*args, **kw)
        tm.stop(name, units)
        return r

    def metric_blocks(self, B):
        """Rows of the f64[blocks, 5] metric-partials buffer step() fills for a minibatch of B rows."""
        return ops.heads_loss_blocks(B, self.H) if self.one_pass_heads else ops.loss_blocks(B)

    @torch.no_grad()
    def step(self, x, act, logp_old, adv, ret, loss_cfg, adv_moments, metric_parts, reducer=None):
        """Forward, loss and backward of one minibatch (module docstring): writes every parameter gradient
        (overwrite) and the f64[metric_blocks(B), 5] partial metric sums.  `reducer` (dist.GradReducer with its
        split at actor.0.weight): bucket 1 = [actor.0 | critic.0, heads] is launched as soon as its last writer is
        queued, bucket 0 = the trunk after the last kernel."""
        m, B, H = self.m, x.shape[0], self.H
        if B > self.R:
            raise ValueError("minibatch larger than the buffers")
        t = self._timed
        H1, H2, Zac, dZ2, dZ1 = self.H1[:B], self.H2[:B], self.Hac[:B], self.dH2[:B], self.dH1[:B]
        W2, ws = m.shared[2].weight, self.gemm_ws
        sfx = "" if H == 256 else f"_h{H}"
        t("linear_tanh_smallk", B, ops.linear_tanh_smallk, x, m.shared[0].weight, m.shared[0].bias, H1)
        t(f"gemm_fwd_{H}_tanh", B, ops.linear_fwd, H1, W2, m.shared[2].bias, H2, True)
        t(f"gemm_fwd_{2 * H}", B, ops.linear_fwd, H2, self.Wac, self.bac, Zac, False)
        heads = (m.actor[2].weight, m.actor[2].bias, m.critic[2].weight, m.critic[2].bias)
        hgrads = (m.actor[2].weight.grad, m.actor[2].bias.grad, m.critic[2].weight.grad, m.critic[2].bias.grad)
        # (multi-rank: bucket 1 of the gradient all-reduce leaves in the middle of step(): its gradients are finalized at once)
        defer = self.defer_finalize and self.one_pass_heads and reducer is None
        if self.one_pass_heads and defer:
            t("heads_loss_fwd_bwd", B, ops.heads_loss_fwd_bwd, Zac, None, *heads, act, logp_old, adv, ret, loss_cfg, adv_moments,
              None, None, None, None, None, metric_parts, self.ws)
            t("gemm_dw_512", B, ops.linear_bwd_weight, Zac, H2, None, ws)
        elif self.one_pass_heads:
            t("heads_loss_fwd_bwd", B, ops.heads_loss_fwd_bwd, Zac, None, *heads, act, logp_old, adv, ret, loss_cfg, adv_moments,
              self.dbac, *hgrads, metric_parts, self.ws)
            t("gemm_dw_512", B, ops.linear_bwd_weight, Zac, H2, self.dWac, ws)
        else:
            logits, value, dl, dv = self.logits[:B], self.value[:B], self.dlogits[:B], self.dvalue[:B]
            # the GEMM added [ba1;bc1]: the heads read pre-activations, recompute their tanh in backward, write dZac in place
            t("heads_fwd_tanh" + sfx, B, ops.heads_fwd_tanh, Zac, heads[0], heads[1], heads[2], heads[3], logits, value, None, False)
            t("ppo_loss_fwd_bwd" + sfx, B, ops.ppo_loss_fwd_bwd, logits, value.view(-1), act, logp_old, adv, ret, loss_cfg,
              adv_moments=adv_moments, dlogits_out=dl, dvalue_out=dv, workspace=metric_parts)
            t("heads_bwd" + sfx, B, ops.heads_bwd, Zac, dl, dv, heads[0], heads[2], Zac, self.dbac, hgrads[0], hgrads[1],
              hgrads[2], hgrads[3], self.ws, True, None)
            t(f"lin_dw_{2 * H}", B, ops.lin_bwd_weight, Zac, None, H2, self.dWac, workspace=self.lin_ws)
        if reducer is not None:
            reducer.launch(1)
        t(f"gemm_dx_{2 * H}_tanhbwd", B, ops.linear_bwd_input, Zac, self.Wac, H2, dZ2)
        if defer:
            t("gemm_dw_256_db", B, ops.linear_bwd_weight, dZ2, H1, None, self.gemm_ws2, None, True)
        elif self.one_pass_heads:
            t("gemm_dw_256_db", B, ops.linear_bwd_weight, dZ2, H1, W2.grad, ws, m.shared[2].bias.grad)
        else:
            t(f"lin_dw_{H}_db", B, ops.lin_bwd_weight, dZ2, None, H1, W2.grad, m.shared[2].bias.grad, workspace=self.lin_ws)
        t(f"gemm_dx_{H}_tanhbwd", B, ops.linear_bwd_input, dZ2, W2, H1, dZ1)
        if defer:
            t("linear_smallk_bwd", B, ops.linear_smallk_bwd, dZ1, None, x, None, None, self.ws2)
            t("update_finalize", B, ops.update_finalize, B, H, self.A, self.D, ws, self.dWac, self.gemm_ws2, W2.grad,
              m.shared[2].bias.grad, self.ws, self.dbac, hgrads[0], hgrads[1], hgrads[2], hgrads[3], self.ws2,
              m.shared[0].weight.grad, m.shared[0].bias.grad)
        else:
            t("linear_smallk_bwd", B, ops.linear_smallk_bwd, dZ1, None, x, m.shared[0].weight.grad, m.shared[0].bias.grad,
              self.ws)
        if reducer is not None:
            reducer.launch(0)
