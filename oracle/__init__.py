"""CPU oracle for the shallow-ntc eval-time hot path.  TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``shallow-ntc_amd/`` imports it, and the product path raises
when the HIP library is missing instead of falling back to this code.

PARITY UNPINNED.  The reference (mandt-lab/shallow-ntc @ 2025-02-27) is ~3.4 kLoC
of Python glue whose arithmetic lives in three un-vendored dependencies that are
not installed here and cannot be fetched (no network):

    tensorflow==2.10.0, tensorflow_compression==2.10.0,
    tensorflow_probability==0.18.0          (reference requirements.txt:6,7,9)

The reference has no tests, no golden tensors and ships no weights, so there is
no tensor-level known answer to pin this restatement against.  What *is* pinned
(tests/test_oracle_pins.py) are the reference's published, checked-in numbers:

  * trainable-parameter counts      results/all_params.csv:2-7
  * FLOPs / pixel at 512x768        results/all_fpp.csv:2-8
  * tensor shapes + "zero input => bias" of the JPEG-like synthesis
                                    notebooks/get_flops.ipynb, vis_syn_filters.ipynb
  * rd_loss = bpp + lambda*mse and psnr = 10 log10(255^2/mse) on the published
    per-image rows                  results/kodak/*-detailed.json
  * scale-table constants           mshyper/models.py:28-34

plus the cross-check that two independent restatements (float64 NumPy in
``ops_np``/``transforms_np`` and float32 PyTorch-CPU in ``torch_ref``) agree.

Modules
  ops_np         float64 NumPy ops (conv / conv-transpose / SignalConv2D / GDN /
                 entropy models / pixel helpers); every function cites the
                 reference call site (file:line) and the dependency semantics it
                 restates.
  transforms_np  layer graphs of common/transforms.py and common/elic.py.
  model_np       mshyper/models.py and factorized/models.py eval forward + SGA loss.
  torch_ref      independent float32 PyTorch-CPU restatement (also the timed CPU
                 baseline, kind "port").
  train_ref      float64 PyTorch-autograd restatement of the training loss (Model.train_step's forward,
                 mshyper/models.py:234-359,375-383, 'unoise' / 'mixedq') as a transforms_np backend: the
                 gradient reference of tests/test_hip_train.py, plus the Keras-Adam / clipnorm arithmetic.
  rans_np        pure-Python restatement of THIS build's rANS wire format and table construction (the
                 reference has no bitstream): pins csrc/rans.hip word for word.
  make_golden    generates tests/golden/*.npz from ops_np/transforms_np/rans_np.
"""
