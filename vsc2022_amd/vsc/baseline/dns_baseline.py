"""Fine-grained (DnS-style) similarity behind the localisation API.

Mirror of `VCSLLocalizationDnS` (/root/reference/vsc/baseline/dns_baseline.py:108-163): the frame x frame
matrix handed to the Temporal Network is produced by a caller-supplied torch similarity network on FINE
(region-level) descriptors, optionally symmetrised and combined with the coarse inner-product similarity by a
geometric mean.  The reference obtains the network from torch hub (`dns_fine_grained_student`); no weights exist
in this environment, so the network is an argument -- any callable `sim_model(query [Lq, ...], ref [Lr, ...]) ->
[Lq, Lr]` with an `fg_type` attribute.

Because `similarity()` is overridden, `localize_all` takes the reference's route of
`vsc2022_amd.vsc.baseline.localization.VCSLLocalization`: one matrix per pair (coarse part computed on the GPU by
libvscmi), alignment of all of them in one `vsc_tn_forward_sim` call, MaxSim score per box.
"""
from typing import Dict, List

import numpy as np
import torch

from vsc2022_amd.vsc.baseline.localization import VCSLLocalizationMaxSim
from vsc2022_amd.vsc.index import VideoFeature
from vsc2022_amd.vsc.metrics import CandidatePair


class VCSLLocalizationDnS(VCSLLocalizationMaxSim):
    def __init__(self, model, queries_fine: List[VideoFeature], refs_fine: List[VideoFeature],
                 queries_coarse: List[VideoFeature], refs_coarse: List[VideoFeature], model_type, device,
                 symmetric: bool = True, geometric_mean: bool = True, **kwargs):
        super().__init__(queries_coarse, refs_coarse, model_type, **kwargs)
        self.queries_fine: Dict[object, VideoFeature] = {v.video_id: v for v in queries_fine}
        self.refs_fine: Dict[object, VideoFeature] = {v.video_id: v for v in refs_fine}
        self.sim_model = model
        self.torch_device = torch.device(device)
        self.symmetric = symmetric
        self.geometric_mean = geometric_mean

    def _rescale_binaries(self, x):
        if "bin" in getattr(self.sim_model, "fg_type", ""):
            x = 2 * x - 1
        return x

    @torch.no_grad()
    def similarity(self, candidate: CandidatePair):
        query = torch.from_numpy(np.asarray(self.queries_fine[candidate.query_id].feature)).to(self.torch_device).float()
        ref = torch.from_numpy(np.asarray(self.refs_fine[candidate.ref_id].feature)).to(self.torch_device).float()
        query, ref = self._rescale_binaries(query), self._rescale_binaries(ref)
        sim = self.sim_model(query, ref)
        if self.symmetric:
            sim = (sim + self.sim_model(ref, query).mT) / 2.0
        sim = (sim / 2.0 + 0.5).cpu().numpy()
        if self.geometric_mean:
            coarse = self._device_similarity(candidate, self.similarity_bias)  # q @ r.T + bias, on the GPU
            sim = np.sqrt(sim.clip(1e-7) * coarse.clip(1e-7))
        return sim
