"""Nearest-centre assignment of generator activations on the MI355X ("next" row (f)-1 of SURVEY.md §8).

Mirrors the prediction half of the reference ``FactorCatalog``
(segmentation/gan_local_edit/factor_catalog.py:47-75): ``predict(X[B,C,H,W]) -> int64 [B,H,W]`` =
``argmin_k ||x - centre_k||^2`` per pixel.  The reference moves the activations to the CPU and builds an
N x K x C difference tensor there (6.7 GB for B=10, K=20 at 256^2); here one HIP kernel reads the activation
once on the device and writes the label map.  Fitting the centres (spherical k-means, sklearn private APIs) is
offline tooling and stays out of scope: centres are supplied as a tensor / loaded from the fitted catalog.
"""
import torch

import sis_hip


class FactorCatalog:
    def __init__(self, k=None, cluster_centers=None, **kwargs):
        self.k = k
        self.cluster_centers = None if cluster_centers is None else torch.as_tensor(cluster_centers, dtype=torch.float32)
        self.annotations = {}

    def pairwise_distance(self, X):
        """X: [N, C] flattened pixels (ptutils.partial_flat layout) -> nearest centre id per row."""
        n, c = X.shape
        return self._assign(X.t().reshape(1, c, n, 1)).reshape(n)

    def _assign(self, X):
        if self.cluster_centers is None:
            raise RuntimeError("FactorCatalog has no cluster centres (fit is offline tooling, load a fitted catalog)")
        sis_hip.require_device(X, "X")
        if self.cluster_centers.device != X.device:
            self.cluster_centers = self.cluster_centers.to(X.device)
        return sis_hip.kmeans_assign(X, self.cluster_centers)

    def predict(self, X):
        batch_size, _, height, width = X.shape
        return self._assign(X).reshape(batch_size, height, width)
