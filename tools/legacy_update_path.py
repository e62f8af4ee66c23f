"""Round-1 update path of the PPO ActorCritic — library GEMMs (torch.mm / addmm -> hipBLASLt) + separate HBM passes —
kept OUT of the product as the A/B baseline of tools/micro_update.py (the product's only path is
gymrl_amd/ppo_net.FusedActorCriticUpdate.step(): hand-written MFMA GEMMs).  Not imported by gymrl_amd."""
import torch

from gymrl_amd import ops
from gymrl_amd.ppo_net import FusedActorCriticUpdate


class LibraryGemmUpdate(FusedActorCriticUpdate):
    """forward() / backward() as they were at the end of round 1."""

    def __init__(self, model, max_rows):
        super().__init__(model, max_rows)
        self._dZac = None
        self._x = None
        self._pre, self._pre_bias = False, None
        self.fused_heads_forward = True
        self.recompute_tanh = True
        self.bwd_in_place = True
        self.recompute_h1 = False
        self.bias_in_gemm = False
        self.overlap_dw = False
        self._side = torch.cuda.Stream(device=self.H1.device)
        if not hasattr(self, "logits"):
            self.logits = torch.empty(self.R, self.A, device=self.H1.device)
            self.value = torch.empty(self.R, 1, device=self.H1.device)

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
