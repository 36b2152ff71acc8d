"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's
UNITER fine-tuning hot path.

Nothing in the product package (``meme_challenge_amd``) imports this.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the reported CPU baseline,
never as the thing measured or shipped.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4).  The oracle is pinned against outputs of the reference itself,
imported in the authoring container with ``apex.FusedLayerNorm`` stubbed by
``torch.nn.LayerNorm`` (Apex's own CPU fallback); the generated vectors are
committed under ``tests/golden/`` together with ``make_golden.py``.
"""
