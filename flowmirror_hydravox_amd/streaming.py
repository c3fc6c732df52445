"""Chunked (streaming) synthesis on top of the three HIP models — SURVEY.md §8(f) N3.

Host-side mirror of `CosyVoice3Model.token2wav` and the `stream=True` branch of `CosyVoice2Model.tts`
(server/model_utils/cosyvoice/cli/model.py:405-430, 316-352): speech tokens arrive from the LM one at a time, every
`token_hop_len` of them (plus `pre_lookahead_len` of look-ahead) the whole prefix is pushed through the flow with the static
chunk mask (`streaming=True`, dit.py:163-164) and `finalize=False`, the new mel frames are appended to the session's mel cache,
HiFT runs over the cached mel with `finalize=False` and only the samples past `speech_offset` are handed out.  All arithmetic is in
libhvx (flow.py / hift.py); this file is bookkeeping only, so it is covered on CPU with stand-in models as well as on the GPU.
"""
import math

import torch


class StreamSession:
    """One utterance being synthesised chunk by chunk.  What the reference keeps in `hift_cache_dict[uuid] = {'mel', 'speech_offset'}`
    (cli/model.py:405-430) lives here as two facts about the session: the mel frames produced so far and how many samples have already
    been handed to the caller.  Every push = flow over the token prefix -> keep only the frames past the prefix already covered ->
    vocode ALL frames so far -> return the samples not yet handed out."""

    def __init__(self, flow, hift, prompt_token, prompt_feat, embedding, device=None):
        self.flow, self.hift = flow, hift
        self.device = device if device is not None else getattr(flow, 'device', 'cpu')
        self.embedding = embedding.to(self.device)
        prompt_token, prompt_feat = prompt_token.to(self.device), prompt_feat.to(self.device)
        # the prompt never changes during an utterance: its flow arguments are fixed here
        self._prompt = {}
        if prompt_token.shape[1] > 0:
            self._prompt = dict(prompt_token=prompt_token, prompt_token_len=torch.tensor([prompt_token.shape[1]], dtype=torch.int32),
                                prompt_feat=prompt_feat, prompt_feat_len=torch.tensor([prompt_feat.shape[1]], dtype=torch.int32))
        self._frames = None            # (1, mel, n) every mel frame of the utterance so far
        self._handed = 0               # samples already returned

    def _fresh_frames(self, token, covered, stream, finalize):
        """mel frames of the tokens past the first `covered` ones (the flow re-runs over the whole prefix, as the reference does)"""
        mel, _ = self.flow.inference(token=token.to(self.device, dtype=torch.int32), token_len=torch.tensor([token.shape[1]], dtype=torch.int32),
                                     embedding=self.embedding, streaming=stream, finalize=finalize, **self._prompt)
        return mel[:, :, covered * self.flow.token_mel_ratio:]

    def _unheard(self, wav):
        out = wav[:, self._handed:]
        self._handed += out.shape[1]
        return out

    def push(self, token, covered, stream=False, finalize=False, speed=1.0):
        """token (1, n): all speech tokens generated so far (plus the look-ahead when not final); `covered`: how many of them earlier
        pushes already turned into frames -> the new samples (1, L)."""
        fresh = self._fresh_frames(token, covered, stream, finalize)
        self._frames = fresh if self._frames is None else torch.concat([self._frames, fresh], dim=2)
        mel = self._frames
        if speed != 1.0:
            if covered != 0 or not finalize:
                raise AssertionError('speed change only support non-stream inference mode')      # cli/model.py:425
            from .ops import resample_linear                                                      # F.interpolate(mode='linear') in libhvx
            mel = resample_linear(mel, int(mel.shape[2] / speed))
        wav, _ = self.hift.inference(speech_feat=mel, finalize=finalize)
        return self._unheard(wav)

    # the reference's name and argument order (cli/model.py:405)
    def token2wav(self, token, token_offset, stream=False, finalize=False, speed=1.0):
        return self.push(token, token_offset, stream=stream, finalize=finalize, speed=speed)


def stream_tts(token_source, flow, hift, prompt_token, prompt_feat, embedding, token_hop_len=25, stream=True, speed=1.0):
    """Generator of waveform pieces (1, L) float32 from an iterable of speech-token ids (e.g. `HvxLLM.inference(...)`).

    The chunk schedule is the reference's (cli/model.py:329-352): the first hop is padded so that prompt + hop is a whole number
    of `token_hop_len` chunks (`prompt_token_pad`), a chunk is synthesised as soon as `hop + pre_lookahead_len` unseen tokens are
    buffered, and whatever remains when the LM stops goes through once more with `finalize=True`.  `token_hop_len` must be
    `static_chunk_size / token_mel_ratio` ("must matching training static_chunk_size", cli/model.py:395-396).
    """
    sess = StreamSession(flow, hift, prompt_token, prompt_feat, embedding)
    tokens = []
    if not stream:
        tokens = [int(t) for t in token_source]
        yield sess.token2wav(torch.tensor(tokens, dtype=torch.int32).unsqueeze(0), 0, stream=False, finalize=True, speed=speed)
        return
    look = flow.pre_lookahead_len
    n_prompt = prompt_token.shape[1]
    prompt_token_pad = int(math.ceil(n_prompt / token_hop_len) * token_hop_len - n_prompt)
    token_offset = 0

    def ready():
        hop = token_hop_len + prompt_token_pad if token_offset == 0 else token_hop_len
        return hop, len(tokens) - token_offset >= hop + look

    for t in token_source:
        tokens.append(int(t))
        hop, ok = ready()
        if ok:
            piece = torch.tensor(tokens[:token_offset + hop + look], dtype=torch.int32).unsqueeze(0)
            wav = sess.token2wav(piece, token_offset, stream=True, finalize=False)
            token_offset += hop
            yield wav
    # the reference's last call leaves `stream` at its default False (cli/model.py:355-361): the final pass uses the full attention mask
    yield sess.token2wav(torch.tensor(tokens, dtype=torch.int32).unsqueeze(0), token_offset, stream=False, finalize=True)
