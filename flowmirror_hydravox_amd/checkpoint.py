"""Checkpoint tooling — SURVEY.md §8(f) N4.

Two things the reference does around its `.pt` files, restated for this build:

* `graft_mtp_heads` — what `scripts/post_process/add_mtp_weights_to_cosyvoice3lm_ckpt.py:131-172` does to an old CosyVoice3LM checkpoint:
  add freshly initialised `mtp_block.{i}.*` decoder layers (one per head) without touching existing entries, then cast every floating
  tensor to bf16.  The script builds `transformers` `Qwen2DecoderLayer(Qwen2Config(hidden_size, num_attention_heads=mtp_head_num,
  num_key_value_heads=mtp_head_num), 0)` after `torch.manual_seed(seed)`; here the same `torch.nn.Linear` / ones tensors are created in
  the module's construction order (q, k, v with bias, o without; gate, up, down without; two RMSNorm gains), which draws the same
  random numbers — the result is bit-identical to the script's (tests/golden/graft_tiny.npz, minted by running the script).
* a packed-weight cache for `ModelManager.load_models` / `load_pt` (`infer_speech_model.py:71-124, 169-184`): the reference re-reads and
  re-casts the `.pt` files on every start and hot swap; this build additionally folds norm gains and weight-norm, permutes rows for the
  fused RoPE and reorders every matrix into MFMA fragment order.  `save_packed` stores those device-ready tensors (safetensors, with
  the configuration, dtype and a layout tag in the metadata); `load_packed` hands them straight to `hvx_*_create`.
"""
import dataclasses
import hashlib
import json
import logging
import os

import torch

# bump when packing.py / the weights[] order documented in include/hvx.h changes
PACK_LAYOUT = 'hvx-pack-5'

QWEN2_DEFAULT_INTERMEDIATE = 22016         # Qwen2Config() default used by the graft script (and by mtp_block, llm_multi_head_v3.py:657-665)

logger = logging.getLogger('hvx.checkpoint')


def graft_mtp_heads(state_dict, head_num=5, mtp_head_num=14, seed=1986, intermediate_size=QWEN2_DEFAULT_INTERMEDIATE):
    """Returns (new_state_dict, n_added).  Existing `mtp_block.*` entries are kept; everything floating ends up bf16 (script :166)."""
    if 'speech_embedding.weight' not in state_dict:
        raise KeyError('checkpoint has no speech_embedding.weight: cannot infer the hidden size')
    if 'llm_decoder.weight' not in state_dict:
        raise KeyError('checkpoint has no llm_decoder.weight')
    vocab, hidden = state_dict['speech_embedding.weight'].shape
    if state_dict['llm_decoder.weight'].shape[0] != vocab:
        raise ValueError('llm_decoder.weight has %d rows, speech_embedding.weight %d' % (state_dict['llm_decoder.weight'].shape[0], vocab))
    if vocab - 200 <= 0:
        raise ValueError('vocab %d leaves no speech tokens (expected speech_token_size + 200)' % vocab)
    head_dim = hidden // mtp_head_num
    attn = mtp_head_num * head_dim
    torch.manual_seed(seed)
    out = dict(state_dict)
    added = 0
    for i in range(head_num):
        # construction order of Qwen2DecoderLayer: self_attn (q, k, v, o), mlp (gate, up, down), input_layernorm, post_attention_layernorm
        lin = [('self_attn.q_proj', hidden, attn, True), ('self_attn.k_proj', hidden, attn, True), ('self_attn.v_proj', hidden, attn, True),
               ('self_attn.o_proj', attn, hidden, False), ('mlp.gate_proj', hidden, intermediate_size, False),
               ('mlp.up_proj', hidden, intermediate_size, False), ('mlp.down_proj', intermediate_size, hidden, False)]
        layer = {}
        for name, fin, fout, bias in lin:
            m = torch.nn.Linear(fin, fout, bias=bias)
            layer[name + '.weight'] = m.weight.detach()
            if bias:
                layer[name + '.bias'] = m.bias.detach()
        layer['input_layernorm.weight'] = torch.ones(hidden)
        layer['post_attention_layernorm.weight'] = torch.ones(hidden)
        for k, v in layer.items():
            key = 'mtp_block.%d.%s' % (i, k)
            if key not in out:
                out[key] = v
                added += 1
    out = {k: (v.to(torch.bfloat16) if torch.is_floating_point(v) else v) for k, v in out.items() if isinstance(v, torch.Tensor)}
    return out, added


def graft_checkpoint_file(src, dst, **kw):
    """file -> file form of the script, including its `{'state_dict': ...}` container handling (:80-92, :168-172)"""
    obj = torch.load(src, map_location='cpu')
    if isinstance(obj, dict) and isinstance(obj.get('state_dict'), dict):
        obj['state_dict'], added = graft_mtp_heads(obj['state_dict'], **kw)
    elif isinstance(obj, dict) and any(isinstance(v, torch.Tensor) for v in obj.values()):
        obj, added = graft_mtp_heads(obj, **kw)
    else:
        raise ValueError("unrecognised checkpoint: expected a state_dict or {'state_dict': state_dict}")
    torch.save(obj, dst)
    return added


# ---- packed-weight cache ----------------------------------------------------------------------------------------------------
def _model_tag(model):
    extra = {k: getattr(model, k) for k in ('max_ctx', 'max_t', 'head_mlp_fp8') if hasattr(model, k)}
    return dict(kind=type(model).__name__, dtype=str(getattr(model, 'dtype', torch.float32)), layout=PACK_LAYOUT,
                cfg=json.dumps(dataclasses.asdict(model.cfg), sort_keys=True), extra=json.dumps(extra, sort_keys=True))


def source_tag(path):
    """identity of a `.pt` file for cache validation: size + mtime + sha256 of its first and last MiB (cheap on multi-GB checkpoints)"""
    st = os.stat(path)
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        h.update(f.read(1 << 20))
        if st.st_size > (2 << 20):
            f.seek(-(1 << 20), os.SEEK_END)
            h.update(f.read(1 << 20))
    return '%d:%d:%s' % (st.st_size, int(st.st_mtime), h.hexdigest()[:32])


def save_packed(model, path, source=''):
    """write the packed tensors of a loaded HvxLLM / HvxFlow / HvxHift; `source` = source_tag() of the `.pt` they came from"""
    from safetensors.torch import save_file
    ws = getattr(model, '_weights', None)
    if not ws:
        raise ValueError('model has no packed weights: load a checkpoint first')
    if hasattr(model, '_cache_tensors'):
        ws = model._cache_tensors()                       # (HvxLLM with fp8 head weights: the codes stand in for the bf16 tensor they expand to)
    meta = _model_tag(model)
    meta.update(n=str(len(ws)), source=source)
    tmp = path + '.tmp'
    save_file({'w%04d' % i: w.detach().contiguous().cpu() for i, w in enumerate(ws)}, tmp, metadata=meta)
    os.replace(tmp, path)
    return path


def read_packed_meta(path):
    from safetensors import safe_open
    with safe_open(path, framework='pt') as f:
        return dict(f.metadata() or {})


def load_packed(model, path, source=None):
    """Fill `model` (constructed without a state dict) from a packed file.  Raises ValueError when the file was written for another
    model class, configuration, dtype, table size or packing layout, or (when `source` is given) from another `.pt`."""
    from safetensors import safe_open
    want = _model_tag(model)
    with safe_open(path, framework='pt') as f:
        meta = dict(f.metadata() or {})
        for k, v in want.items():
            if meta.get(k) != v:
                raise ValueError('packed file %s: %s is %r, this model needs %r' % (path, k, meta.get(k), v))
        if source is not None and meta.get('source') != source:
            raise ValueError('packed file %s was made from another checkpoint' % path)
        n = int(meta.get('n', -1))
        keys = set(f.keys())
        if n <= 0 or any('w%04d' % i not in keys for i in range(n)):
            raise ValueError('packed file %s is incomplete (%d tensors announced, %d present)' % (path, n, len(keys)))
        ws = [f.get_tensor('w%04d' % i) for i in range(n)]
    return model.load_packed(ws)


def load_or_pack(model, pt_path, cache_dir, loader):
    """`loader(pt_path) -> state_dict`.  Uses `<cache_dir>/<basename>.<kind>.hvxpack` when it matches the `.pt`, the model and the layout;
    otherwise packs from the `.pt` and (re)writes the cache.  Returns 'cache' or 'packed'."""
    src = source_tag(pt_path)
    os.makedirs(cache_dir, exist_ok=True)
    cache = os.path.join(cache_dir, '%s.%s.hvxpack' % (os.path.basename(pt_path), type(model).__name__))
    if os.path.exists(cache):
        try:
            load_packed(model, cache, source=src)
            return 'cache'
        except Exception as e:          # stale / truncated / foreign file, or the native layer rejecting its contents: repack from the .pt
            logger.warning('packed cache %s unusable (%s: %s); repacking from %s', cache, type(e).__name__, e, pt_path)
    model.load_state_dict(loader(pt_path))
    try:
        save_packed(model, cache, source=src)
    except Exception as e:              # a read-only or full cache directory must not fail the load
        logger.warning('could not write packed cache %s (%s: %s)', cache, type(e).__name__, e)
    return 'packed'
