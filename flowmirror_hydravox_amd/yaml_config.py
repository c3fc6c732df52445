"""Model dimensions from `<model_dir>/hydravox.yaml` — the file the reference builds its modules from
(server/model_utils/infer_speech_model.py:59-62: `load_hyperpyyaml(f, overrides={'qwen_pretrain_path': ...})`).

HyperPyYAML (not part of this build) instantiates Python objects from `!new:module.Class` nodes; this reader only needs the plain
numbers those nodes carry.  It parses the file with PyYAML, keeps every tagged node (`!new:`, `!name:`, `!apply:`) as a dict of its
keyword arguments plus `__tag__`, resolves `!ref <key>` / `!ref <a> * <b>` references against the top level, and maps the constructor
arguments of the reference classes onto HvxConfig by name:

    llm   : CosyVoice3LM(llm_input_size, llm_output_size, speech_token_size, head_num, mtp_head_num, sampling=ras_sampling(...))
            (cosyvoice/llm/llm_multi_head_v3.py:621-640); the Qwen2 backbone sizes come from `<model_dir>/CosyVoice-BlankEN/config.json`
            (the HF config `Qwen2Encoder` loads, llm_multi_head_v3.py:231-236) when that file exists
    flow  : CausalMaskedDiffWithDiT(output_size, spk_embed_dim, vocab_size, token_mel_ratio, pre_lookahead_len, pre_lookahead_layer,
            decoder=CausalConditionalCFM(cfm_params{inference_cfg_rate}, estimator=DiT(dim, depth, heads, dim_head, ff_mult, mel_dim,
            static_chunk_size)))            (cosyvoice/flow/flow.py:278-312, flow/flow_matching.py:197, flow/DiT/dit.py:104-120)
    hift  : CausalHiFTGenerator(in_channels, base_channels, nb_harmonics, sampling_rate, nsf_*, upsample_*, istft_params,
            resblock_*, source_resblock_*, lrelu_slope, audio_limit, conv_pre_look_right, f0_predictor(cond_channels))
            (cosyvoice/hifigan/generator.py:573-600, hifigan/f0_predictor.py:60-66)

Anything the HIP path cannot run (e.g. a DiT head size other than 64) raises ValueError naming the field.
"""
import ast
import json
import operator
import os
import re

from .config import FlowConfig, HiftConfig, HvxConfig, LLMConfig

_REF = re.compile(r'<([A-Za-z_][\w.]*)>')
_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv, ast.FloorDiv: operator.floordiv,
        ast.Mod: operator.mod, ast.Pow: operator.pow, ast.USub: operator.neg, ast.UAdd: operator.pos}


class _Ref(str):
    """the raw text of a `!ref` scalar"""


def _loader():
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def tagged(loader, suffix, node):
        tag = node.tag
        if isinstance(node, yaml.MappingNode):
            d = loader.construct_mapping(node, deep=True)
            d['__tag__'] = tag
            return d
        if isinstance(node, yaml.SequenceNode):
            return {'__tag__': tag, '__args__': loader.construct_sequence(node, deep=True)}
        v = loader.construct_scalar(node)
        if tag.startswith('!ref'):
            return _Ref(v)
        return {'__tag__': tag, '__value__': v}

    Loader.add_multi_constructor('!', tagged)
    return Loader


def _arith(expr):
    """value of a plain arithmetic expression over numbers (what is left of a `!ref` once its <keys> are substituted)"""
    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return n.value
        if isinstance(n, ast.BinOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.operand))
        raise ValueError('unsupported expression in !ref: %r' % expr)
    return ev(ast.parse(expr, mode='eval'))


def _lookup(root, dotted):
    cur = root
    for part in dotted.split('.'):
        if isinstance(cur, dict) and part in cur:
            cur = cur[part]
        elif isinstance(cur, list) and part.isdigit() and int(part) < len(cur):
            cur = cur[int(part)]
        else:
            raise ValueError('hydravox.yaml: !ref <%s> does not resolve' % dotted)
    return cur


def _resolve(node, root, depth=0):
    if depth > 32:
        raise ValueError('hydravox.yaml: reference cycle')
    if isinstance(node, _Ref):
        text = str(node).strip()
        whole = _REF.fullmatch(text)
        if whole:                                              # a pure reference keeps the referenced node (number, list, mapping)
            return _resolve(_lookup(root, whole.group(1)), root, depth + 1)
        def sub(m):
            v = _resolve(_lookup(root, m.group(1)), root, depth + 1)
            if isinstance(v, bool) or not isinstance(v, (int, float, str)):
                raise ValueError('hydravox.yaml: <%s> used inside an expression is not a scalar' % m.group(1))
            return repr(v) if not isinstance(v, str) else v
        flat = _REF.sub(sub, text)
        try:
            return _arith(flat)
        except (ValueError, SyntaxError):
            return flat                                        # string interpolation (paths)
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    return node


def load_hyperpyyaml_plain(text, overrides=None):
    """-> the document as plain dicts / lists / scalars with references resolved; tagged nodes carry '__tag__'"""
    import yaml
    doc = yaml.load(text, Loader=_loader()) or {}
    if overrides:
        doc.update(overrides)
    return _resolve(doc, doc)


def _node(d, key, what):
    v = d.get(key)
    if v is None:
        return {}
    if not isinstance(v, dict):
        raise ValueError('hydravox.yaml: %s.%s is not a mapping' % (what, key))
    return v


def _take(d, key, default, cast=None):
    v = d.get(key, default)
    if v is None:
        return default
    return cast(v) if cast else v


def config_from_yaml_doc(doc, qwen_cfg=None):
    """plain document (load_hyperpyyaml_plain) [+ the HF Qwen2 config dict] -> (HvxConfig, extras); extras holds what is not a dimension:
    the yaml's sampling defaults and inference_head_num"""
    L0, F0, H0 = LLMConfig(), FlowConfig(), HiftConfig()
    llm_n, flow_n, hift_n = _node(doc, 'llm', 'root'), _node(doc, 'flow', 'root'), _node(doc, 'hift', 'root')

    # ---- llm ---------------------------------------------------------------------------------------------------------
    hidden = int(_take(llm_n, 'llm_input_size', doc.get('llm_input_size', L0.hidden)))
    if int(_take(llm_n, 'llm_output_size', doc.get('llm_output_size', hidden))) != hidden:
        raise ValueError('hydravox.yaml: llm_output_size != llm_input_size is not supported (the speech embedding feeds the backbone directly)')
    q = qwen_cfg or {}
    heads = int(q.get('num_attention_heads', L0.q_heads))
    if q and int(q.get('hidden_size', hidden)) != hidden:
        raise ValueError('hydravox.yaml: llm_input_size %d != hidden_size %d of the Qwen2 config' % (hidden, int(q['hidden_size'])))
    head_dim = hidden // heads
    if head_dim != 64:
        raise ValueError('Qwen2 head_dim %d: the HIP attention kernels are built for 64' % head_dim)
    llm = LLMConfig(hidden=hidden, layers=int(q.get('num_hidden_layers', L0.layers)), q_heads=heads,
                    kv_heads=int(q.get('num_key_value_heads', L0.kv_heads)), head_dim=head_dim,
                    inter=int(q.get('intermediate_size', L0.inter)), rope_theta=float(q.get('rope_theta', L0.rope_theta)),
                    rms_eps=float(q.get('rms_norm_eps', L0.rms_eps)), text_vocab=int(q.get('vocab_size', L0.text_vocab)),
                    speech_tokens=int(_take(llm_n, 'speech_token_size', L0.speech_tokens)), extra_tokens=L0.extra_tokens,
                    head_num=int(_take(llm_n, 'head_num', L0.head_num)), mtp_heads=int(_take(llm_n, 'mtp_head_num', L0.mtp_heads)),
                    mtp_inter=L0.mtp_inter, mtp_rms_eps=L0.mtp_rms_eps)
    samp = _node(llm_n, 'sampling', 'llm')
    extras = {'sampling': {k: samp[k] for k in ('top_p', 'top_k', 'win_size', 'tau_r') if k in samp},
              'inference_head_num': int(_take(llm_n, 'inference_head_num', llm.head_num))}

    # ---- flow --------------------------------------------------------------------------------------------------------
    dec = _node(flow_n, 'decoder', 'flow')
    est = _node(dec, 'estimator', 'flow.decoder')
    pla = _node(flow_n, 'pre_lookahead_layer', 'flow')
    cfm = dec.get('cfm_params') or {}
    if isinstance(cfm, dict) and '__args__' in cfm and cfm['__args__'] and isinstance(cfm['__args__'][0], dict):
        cfm = cfm['__args__'][0]                               # `!new:omegaconf.DictConfig` with a `content:` / positional mapping
    if isinstance(cfm, dict) and isinstance(cfm.get('content'), dict):
        cfm = cfm['content']
    mel = int(_take(flow_n, 'output_size', F0.mel))
    dim_head = int(_take(est, 'dim_head', F0.head_dim))
    if dim_head != 64:
        raise ValueError('DiT dim_head %d: the HIP attention kernels are built for 64' % dim_head)
    dim, heads_f = int(_take(est, 'dim', F0.dim)), int(_take(est, 'heads', F0.heads))
    if dim != heads_f * dim_head:
        raise ValueError('DiT dim %d != heads %d x dim_head %d' % (dim, heads_f, dim_head))
    for k in ('mel_dim', 'mu_dim', 'spk_dim', 'out_channels'):
        if est.get(k) not in (None, mel):
            raise ValueError('DiT %s = %s differs from the mel width %d (the estimator input is cat[x, cond, mu, spks] of equal widths)' % (k, est[k], mel))
    if str(cfm.get('solver', 'euler')) != 'euler' or str(cfm.get('t_scheduler', 'cosine')) != 'cosine':
        raise ValueError('CFM solver / t_scheduler other than euler / cosine are not supported')
    flow = FlowConfig(vocab=int(_take(flow_n, 'vocab_size', F0.vocab)), mel=mel,
                      spk_embed_dim=int(_take(flow_n, 'spk_embed_dim', doc.get('spk_embed_dim', F0.spk_embed_dim))),
                      token_mel_ratio=int(_take(flow_n, 'token_mel_ratio', doc.get('token_mel_ratio', F0.token_mel_ratio))),
                      pre_lookahead_len=int(_take(pla, 'pre_lookahead_len', _take(flow_n, 'pre_lookahead_len', F0.pre_lookahead_len))),
                      pre_lookahead_channels=int(_take(pla, 'channels', F0.pre_lookahead_channels)),
                      n_timesteps=F0.n_timesteps, cfg_rate=float(cfm.get('inference_cfg_rate', F0.cfg_rate)), noise_frames=F0.noise_frames,
                      dim=dim, depth=int(_take(est, 'depth', F0.depth)), heads=heads_f, head_dim=dim_head,
                      ff_mult=int(_take(est, 'ff_mult', F0.ff_mult)), conv_kernel=F0.conv_kernel, conv_groups=F0.conv_groups,
                      time_freq_dim=F0.time_freq_dim, static_chunk_size=int(_take(est, 'static_chunk_size', F0.static_chunk_size)))

    # ---- hift --------------------------------------------------------------------------------------------------------
    istft = hift_n.get('istft_params') or {}
    f0p = _node(hift_n, 'f0_predictor', 'hift')
    hift = HiftConfig(mel=int(_take(hift_n, 'in_channels', H0.mel)), base_channels=int(_take(hift_n, 'base_channels', H0.base_channels)),
                      nb_harmonics=int(_take(hift_n, 'nb_harmonics', H0.nb_harmonics)),
                      sampling_rate=int(_take(hift_n, 'sampling_rate', doc.get('sample_rate', H0.sampling_rate))),
                      nsf_alpha=float(_take(hift_n, 'nsf_alpha', H0.nsf_alpha)), nsf_sigma=float(_take(hift_n, 'nsf_sigma', H0.nsf_sigma)),
                      nsf_voiced_threshold=float(_take(hift_n, 'nsf_voiced_threshold', H0.nsf_voiced_threshold)),
                      upsample_rates=[int(v) for v in _take(hift_n, 'upsample_rates', H0.upsample_rates)],
                      upsample_kernel_sizes=[int(v) for v in _take(hift_n, 'upsample_kernel_sizes', H0.upsample_kernel_sizes)],
                      n_fft=int(istft.get('n_fft', H0.n_fft)), hop=int(istft.get('hop_len', H0.hop)),
                      resblock_kernel_sizes=[int(v) for v in _take(hift_n, 'resblock_kernel_sizes', H0.resblock_kernel_sizes)],
                      resblock_dilations=[[int(x) for x in v] for v in _take(hift_n, 'resblock_dilation_sizes', H0.resblock_dilations)],
                      source_resblock_kernel_sizes=[int(v) for v in _take(hift_n, 'source_resblock_kernel_sizes', H0.source_resblock_kernel_sizes)],
                      source_resblock_dilations=[[int(x) for x in v] for v in _take(hift_n, 'source_resblock_dilation_sizes', H0.source_resblock_dilations)],
                      lrelu_slope=float(_take(hift_n, 'lrelu_slope', H0.lrelu_slope)), audio_limit=float(_take(hift_n, 'audio_limit', H0.audio_limit)),
                      conv_pre_look_right=int(_take(hift_n, 'conv_pre_look_right', H0.conv_pre_look_right)),
                      f0_channels=int(_take(f0p, 'cond_channels', H0.f0_channels)), noise_seconds=H0.noise_seconds)
    if hift.mel != flow.mel:
        raise ValueError('hift in_channels %d != flow output_size %d' % (hift.mel, flow.mel))
    if len(hift.upsample_rates) != len(hift.upsample_kernel_sizes) or len(hift.source_resblock_kernel_sizes) != len(hift.upsample_rates):
        raise ValueError('hift: upsample_rates / upsample_kernel_sizes / source_resblock_kernel_sizes must have one entry per stage')
    return HvxConfig(llm=llm, flow=flow, hift=hift, sample_rate=int(doc.get('sample_rate', hift.sampling_rate))), extras


def config_from_model_dir(model_dir):
    """<model_dir>/hydravox.yaml (+ CosyVoice-BlankEN/config.json) -> (HvxConfig, extras), or None when there is no yaml"""
    path = os.path.join(model_dir, 'hydravox.yaml')
    if not os.path.exists(path):
        return None
    with open(path, 'r') as f:
        doc = load_hyperpyyaml_plain(f.read(), overrides={'qwen_pretrain_path': os.path.join(model_dir, 'CosyVoice-BlankEN')})
    qwen = None
    qpath = os.path.join(model_dir, 'CosyVoice-BlankEN', 'config.json')
    if os.path.exists(qpath):
        with open(qpath) as f:
            qwen = json.load(f)
    return config_from_yaml_doc(doc, qwen)
