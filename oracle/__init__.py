"""CPU oracle for the text-conditioned separation hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package - as the checker,
never as the thing measured or shipped.  Nothing under lass_amd/ imports it; the product path fails loudly when the
HIP extension is missing.

It is a from-scratch restatement (torch-CPU / numpy, functional, explicit weight dict) of the reference algorithm:
    oracle.stft      <- torchlibrosa 0.1.0 STFT / ISTFT / magphase semantics (third party, absent; see header there)
    oracle.resunet   <- /root/reference/models/resunet.py, models/base.py
    oracle.metrics   <- /root/reference/utils.py:148-200
    oracle.evaluator <- /root/reference/dcase_evaluator.py:27-145

Pinning status (see DESIGN.md "Oracle"):
  * FiLM / ResUNet30 / mask arithmetic: PINNED - golden vectors in tests/golden/ were produced in the build
    container by running the reference's own models/resunet.py (tools/gen_golden.py).
  * STFT / iSTFT numerics at the torchlibrosa boundary: "parity unpinned" against torchlibrosa itself (library absent,
    no fixtures in the reference); pinned instead to torch.stft/torch.istft equivalence and the round-trip identity.
  * SDR / SI-SDR: pinned by closed-form cases (the reference ships no vectors).
"""
