"""CPU oracle for the ipr-gan G+D training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ipr-gan_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

What it is: a plain-PyTorch (CPU, fp32, NCHW) restatement of the reference's
networks, step choreography and sign-loss watermark, written from the
reference's behaviour (each function cites the reference file:line it
follows).  The arithmetic itself (conv, norm, Adam, spectral-norm power
iteration) lives in the third-party dependency torch (reference pins
torch==1.8.0, requirements.txt:10); here torch 2.10 CPU ops stand in for it.

Parity pin: ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` (possible only in the build container) and writes golden
vectors to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against them.  Not pinned (stated in DESIGN.md): VGG19 pretrained
weights and pytorch-msssim SSIM, which are absent from the image.
"""
