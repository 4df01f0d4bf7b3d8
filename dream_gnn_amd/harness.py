"""Harness: the training / evaluation step *structure* of the reference, restated so the HIP
path can be driven end to end on a box with neither DGL nor the reference (scope row H).

Mirrors, by line:
  * the augmentation defaults of ``train.py:238,267`` — ``edge_dropout`` on the encoder graph
    (augmentation.py:13-89) and on the four sparse adjacencies (:92-124), ``feature_noise`` on
    the four feature matrices (:208-241, scales 0.05 / sim 0.05 via augmentation.py:473-476);
  * one optimisation step ``train.py:280-300`` — forward, ``BCEWithLogits + beta * (common_loss
    drug + common_loss disease)``, backward, ``clip_grad_norm_``, optimiser step;
  * the metric lines of ``evaluation.py:55-65`` — trapezoid AUROC and trapezoid PR-AUC
    ``auc(recall, precision)`` (not average precision).
Nothing here is on the hot path; it only calls it.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import graph as G
from .model import common_loss


def augment(batch: Dict, edge_dropout_rate: float = 0.1, feature_noise_scale: float = 0.05,
            sim_noise_scale: float = 0.05, generator: Optional[torch.Generator] = None,
            selection: Optional[str] = None) -> Dict:
    """``augment_graph_data(..., ['edge_dropout', 'feature_noise'])`` — augmentation.py:402-489.
    The decoder graph is not augmented (train.py:270).  ``selection``: how the edge subsets are drawn
    (``graph.random_edge_dropout``); ``"select_device"`` draws their seeds on the device — no host value, no
    synchronisation: the form :class:`CapturedTrainStep` records."""
    out = dict(batch)
    out["enc_graph"] = G.random_edge_dropout(batch["enc_graph"], edge_dropout_rate, generator, selection=selection)
    # masked views of the cached CSRs (GraphConvolution accepts them; the reference's sparse tensors are
    # available from G.random_edge_dropout_sparse), all four subsets selected by one series of launches
    keys = [k for k in ("drug_graph", "disease_graph", "drug_feature_graph", "disease_feature_graph")
            if batch.get(k) is not None]
    for k, view in zip(keys, G.random_edge_dropout_sparse_views([batch[k] for k in keys], edge_dropout_rate, generator,
                                                                selection=selection)):
        out[k] = view
    for k, scale in (("drug_feat", feature_noise_scale), ("disease_feat", feature_noise_scale),
                     ("drug_sim_feat", sim_noise_scale), ("disease_sim_feat", sim_noise_scale)):
        if batch.get(k) is not None:
            x = batch[k]
            if generator is not None and generator.device.type == "cpu" and x.device.type != "cpu":
                noise = torch.randn(x.shape, dtype=x.dtype, generator=generator).to(x.device)  # host draw, as graph._randperm
            else:
                noise = torch.randn(x.shape, dtype=x.dtype, device=x.device, generator=generator)
            out[k] = x + noise * scale
    return out


def forward_loss(net, batch: Dict, labels: torch.Tensor, beta: float):
    """train.py:280-294."""
    pred, drug_out, drug_sim_out, dis_out, dis_sim_out = net(
        batch["enc_graph"], batch["dec_graph"], batch["drug_graph"], batch["drug_sim_feat"], batch["drug_feat"],
        batch["disease_graph"], batch["disease_sim_feat"], batch["disease_feat"],
        batch.get("drug_feature_graph"), batch.get("disease_feature_graph"), False)
    pred = pred.squeeze(-1)
    rel = nn.functional.binary_cross_entropy_with_logits(pred, labels)
    loss = rel + beta * (common_loss(drug_out, drug_sim_out) + common_loss(dis_out, dis_sim_out))
    return loss, pred


def train_step(net, optimizer, batch: Dict, labels: torch.Tensor, beta: float = 0.1, grad_clip: float = 1.0,
               do_augment: bool = True, generator: Optional[torch.Generator] = None,
               selection: Optional[str] = None) -> torch.Tensor:
    """One iteration of train.py:249-300.  Returns the (detached) total loss.  ``generator`` / ``selection``: where the
    augmentation's draws come from (``augment``)."""
    net.train()
    step_batch = augment(batch, generator=generator, selection=selection) if do_augment else batch
    loss, _ = forward_loss(net, step_batch, labels, beta)
    optimizer.zero_grad()
    loss.backward()
    nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
    optimizer.step()
    return loss.detach()


class CapturedTrainStep:
    """One training iteration (train.py:249-300: per-step augmentation, forward, loss, backward, clip, Adam) recorded
    once as a HIP graph and replayed.  The real datasets are launch-bound on an MI355X (an lrssl-shaped step is ~600
    small kernels: 7-9 ms eager, host-bound; ~6.5 ms of device time): a replay issues them with one call.

    What makes the step recordable is that nothing in it needs the host: the edge dropout is an 8-word description
    selected on the device (``selection="select_device"``: the subsets' seeds come from the device RNG, so every replay
    drops different edges, exactly ``max(1, int(E (1 - p)))`` kept each time), no graph is rebuilt, the layouts are
    cached on the un-dropped parents, dropout masks and feature noise use torch's graph-aware device generator, and the
    optimizer must be ``capturable`` (``torch.optim.Adam(..., capturable=True)``).

    ``step = CapturedTrainStep(net, opt, batch, labels); loss = step()`` — ``loss`` is a device tensor that the next
    replay overwrites.  The model's parameters, the optimizer state and ``batch`` / ``labels`` are captured by
    address: update them in place (``labels.copy_(...)``), never rebind them."""

    def __init__(self, net, optimizer, batch: Dict, labels: torch.Tensor, beta: float = 0.1, grad_clip: float = 1.0,
                 do_augment: bool = True, warmup: int = 3):
        """NOTE: constructing the object TRAINS: ``warmup`` (>= 1, default 3) real iterations run eagerly on a side stream
        before the recording (layouts, plans and autotuned library kernels must exist before a capture), and the
        recording itself executes nothing.  Parameters, Adam moments / step count and the RNG streams advance by
        ``warmup`` steps: N replays after construction leave the model where ``warmup + N`` eager steps would."""
        if not labels.is_cuda:
            raise RuntimeError("CapturedTrainStep records a HIP graph: the model and its inputs must be on the GPU")
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise RuntimeError("the optimizer must be constructed with capturable=True to be recorded in a HIP graph")
        self.net, self.optimizer = net, optimizer

        def step():
            net.train()
            step_batch = augment(batch, selection="select_device") if do_augment else batch
            loss, _ = forward_loss(net, step_batch, labels, beta)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
            optimizer.step()
            return loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # lazily built layouts, plans and autotuned library kernels exist before the recording
            for _ in range(max(1, warmup)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def __call__(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss


def _binary_clf_curve(y_true: np.ndarray, y_score: np.ndarray):
    order = np.argsort(-y_score, kind="mergesort")
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    return fps, tps


def _trapz(y, x):
    return float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2.0))


def auroc_aupr(y_true, y_score):
    """``metrics.auc(*roc_curve)`` and ``metrics.auc(recall, precision)`` — evaluation.py:60-65."""
    y_true = np.asarray(y_true, dtype=np.float64).reshape(-1)
    y_score = np.asarray(y_score, dtype=np.float64).reshape(-1)
    fps, tps = _binary_clf_curve(y_true, y_score)
    # ROC: curve starts at (0, 0); dropping collinear points (sklearn's drop_intermediate) does
    # not change the trapezoid area
    fpr = np.r_[0.0, fps] / fps[-1]
    tpr = np.r_[0.0, tps] / tps[-1]
    auroc = _trapz(tpr, fpr)
    # PR: one point per distinct threshold, plus the (recall 0, precision 1) end point
    ps = tps + fps
    precision = np.divide(tps, ps, out=np.zeros_like(tps), where=ps != 0)
    recall = tps / tps[-1]
    precision = np.r_[precision[::-1], 1.0]
    recall = np.r_[recall[::-1], 0.0]
    aupr = -_trapz(precision, recall)  # recall is decreasing
    return auroc, aupr


@torch.no_grad()
def evaluate(net, batch: Dict, labels: torch.Tensor):
    """evaluation.py:44-65: eval-mode forward on the un-augmented graphs, then the two areas."""
    net.eval()
    pred, *_ = net(batch["enc_graph"], batch["dec_graph"], batch["drug_graph"], batch["drug_sim_feat"],
                   batch["drug_feat"], batch["disease_graph"], batch["disease_sim_feat"], batch["disease_feat"],
                   batch.get("drug_feature_graph"), batch.get("disease_feature_graph"))
    return auroc_aupr(labels.cpu().numpy(), pred.view(-1).cpu().numpy())
