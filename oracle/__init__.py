"""CPU oracle for the STMoGen sampling hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker -- never as the thing measured or shipped.  The
product path (``motioncraft_amd``) never imports this package and fails loudly
when its HIP library is missing.

Contents
--------
``tutel_restated.py``  restatement of the un-vendored third-party dependency
                       ``tutel.moe.moe_layer`` (microsoft/tutel, version
                       unpinned by the reference).  **PARITY UNPINNED**: the
                       reference holds no test or golden vector at that call
                       site (reference ``mogen/models/attentions/st_attention.py:28-45``)
                       and tutel is not installed here, so this file *is* the
                       spec the build is checked against.
``stmogen_oracle.py``  pure torch-CPU restatement of the whole per-step path
                       (no reference imports); travels to the GPU box.
``ref_shim.py``        loads the reference's own hot-path modules from
                       ``/root/reference`` by file path with stubs for
                       mmcv/clip/tutel.  Works only in the build container;
                       used by ``tests/golden/make_golden.py`` to pin
                       ``stmogen_oracle`` and to generate the golden fixtures.
``weights.py``         deterministic (seed, state-dict key) -> tensor init so
                       weights never need to be committed.
"""
