"""oracle/ — CPU restatement of the reference algorithm for the HydraVox hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`flowmirror_hydravox_amd/`) may import,
call, link or execute anything in this directory; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and there only as the checker / reported baseline.

The reference (jingzhunxue/FlowMirror_HydraVox) is pure Python on PyTorch, so the restatement is
plain functional fp32 PyTorch-on-CPU (no nn.Module, weights in a flat dict that uses the
reference's own state_dict key names).  Each function cites the reference file:line it follows.
`llm_ref` / `flow_ref` also have a bf16-faithful mode (`emu=True`): the same functions with every
operand rounded to bf16 where the product's bf16 path holds it in bf16 and fp32 accumulation — the
yardstick for the production dtype (tests/test_gpu_cv3w.py), never the parity target.

Pinning status: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself, produced in the
build container by importing the reference modules with seeded random weights
(`tests/golden/make_golden.py`, fixtures under `tests/golden/*.npz`).  Third-party arithmetic that
is absent from /root/reference (HF transformers Qwen2 layers, pinned transformers==4.40.1;
x_transformers==2.12.2 rotary embedding) is restated from the published algorithm and, for Qwen2,
checked against the installed transformers build; the x_transformers rotary restatement has no
reference-side check available here -> "parity unpinned" for that one sub-step (DiT RoPE).
"""
