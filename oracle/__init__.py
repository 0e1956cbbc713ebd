"""CPU oracle for the eps_theta(x_t, h_t, t) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the CPU baseline.
The product path (``diffusion_model_amd``) never imports this package and
fails loudly when its HIP extension is missing.

The oracle is a plain torch-CPU (fp32) restatement of the reference's
algorithm.  Each function cites the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * schedule / diffusion-step / remove_mean / GammaNetwork / SpectrumCompressor
    functions are pinned by golden vectors produced by importing the reference
    modules in the build container (tests/golden/make_golden.py).
  * EGCL / EquivariantGNN are pinned by golden vectors produced by executing
    the reference's EquivariantGraphNeuralNetwork.py verbatim against a
    ~25-line stand-in for torch_geometric.nn.MessagePassing (torch_geometric
    itself is un-vendored, un-pinned and absent offline; the reference holds no
    tests or fixtures for that boundary).  The gather/scatter semantics of the
    stand-in follow PyG's documented flow='target_to_source', aggr='sum'.
  * RDF / Si-O-Si statistics: parity unpinned by execution (the reference files
    import wandb at module top level); restated from the text and pinned by
    hand-computed geometries only.
"""
