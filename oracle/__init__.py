"""CPU oracle for the synthesis-in-style hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``synthesis-in-style_amd/``) never imports this package and fails loudly when
its HIP library is missing.

Contents
--------
ops_ref.py        torch-CPU restatement of the two native ops of the reference
                  (``upfirdn2d``, ``fused_bias_act``) in the formulation of
                  ``stylegan_code_finder/networks/stylegan2/op/upfirdn2d.py:152-186``
                  and ``.../fused_bias_act_kernel.cu:25-47``.
ops_c.c / c_ops.py  an independent, index-level plain-C restatement of the two
                  CUDA kernels (``upfirdn2d_kernel.cu:83-134``,
                  ``fused_bias_act_kernel.cu:18-49``), used to cross-check
                  ops_ref.py.
stylegan2_ref.py  functional restatement of ``Generator.forward``
                  (``networks/stylegan2/model.py:479-561``) over the 135-key
                  ``g_ema`` state_dict, using per-sample weight materialisation
                  and grouped convolutions exactly as ``model.py:237-278``.
segmentation_ref.py  stock-torch restatement of the EMANet / TransUNet training
                  step (``updater/segmentation_updater.py:47-106``).
load_reference.py imports the *unmodified* reference modules by file path with
                  ``.op`` replaced by ops_ref (only where ``/root/reference``
                  exists: this container; never on the GPU box).

Pinning: the reference's own tests hold no vectors for this path,
so the oracle is pinned by (1) ops_ref vs ops_c agreement, (2) stylegan2_ref vs
the imported reference ``model.py`` run here, (3) the golden fixtures under
``tests/golden`` produced by ``tests/golden/make_golden.py`` from the imported
reference.
"""
