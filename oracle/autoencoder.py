"""
ORACLE (test infrastructure only -- never imported by the product path).

Functional CPU restatement (torch float32/float64 on the CPU, stock ``torch.nn.functional``
ops + autograd) of the Timbre-Trap autoencoder in reference
``timbre_trap/framework/modules.py``.  Weights are taken from a plain dict that uses the
reference ``state_dict`` key names, so the same dict drives the reference (when the golden
fixtures are generated), this oracle and the HIP path.

Pinned against the imported reference by tests/golden/*.npz (generator:
tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""

import math

import torch
import torch.nn.functional as F


def channels_for(model_complexity):
    """modules.py:417-424 (encoder order; the decoder uses it reversed, :507-514)."""
    return tuple(round(c * 2 ** (model_complexity - 1)) for c in (2, 4, 8, 16, 32))


def default_latent(model_complexity):
    """modules.py:426-428."""
    return 32 * 2 ** (model_complexity - 1)


def embedding_sizes(feature_size):
    """modules.py:440-444 / :520-531 : heights per level and decoder output paddings."""
    sizes, padding = [feature_size], []
    e = feature_size
    for _ in range(4):
        padding.append(e % 2)
        e = e // 2 - 1
        sizes.append(e)
    padding.reverse()
    return sizes, padding


def state_dict_shapes(feature_size=540, latent_size=None, model_complexity=1, skip_connections=False):
    """Key -> shape of the reference state_dict, in the reference's registration order."""
    ch = channels_for(model_complexity)
    lat = default_latent(model_complexity) if latent_size is None else latent_size
    sizes, _ = embedding_sizes(feature_size)
    shapes = {}
    if skip_connections:
        shapes['skip_weights'] = (5,)

    def res(prefix, c):
        for i in (1, 2, 3):
            shapes[f'{prefix}.block{i}.conv1.0.weight'] = (c, c, 3, 3)
            shapes[f'{prefix}.block{i}.conv1.0.bias'] = (c,)
            shapes[f'{prefix}.block{i}.conv2.0.weight'] = (c, c, 1, 1)
            shapes[f'{prefix}.block{i}.conv2.0.bias'] = (c,)

    shapes['encoder.convin.0.weight'] = (ch[0], 2, 3, 3)
    shapes['encoder.convin.0.bias'] = (ch[0],)
    for i in range(4):
        res(f'encoder.block{i + 1}', ch[i])
        shapes[f'encoder.block{i + 1}.sconv.0.weight'] = (ch[i + 1], ch[i], 4, 1)
        shapes[f'encoder.block{i + 1}.sconv.0.bias'] = (ch[i + 1],)
    shapes['encoder.convlat.weight'] = (lat, ch[4], sizes[4], 1)
    shapes['encoder.convlat.bias'] = (lat,)

    dch = ch[::-1]
    shapes['decoder.convin.0.weight'] = (lat + 1, dch[0], sizes[4], 1)
    shapes['decoder.convin.0.bias'] = (dch[0],)
    for i in range(4):
        shapes[f'decoder.block{i + 1}.tconv.0.weight'] = (dch[i], dch[i + 1], 4, 1)
        shapes[f'decoder.block{i + 1}.tconv.0.bias'] = (dch[i + 1],)
        res(f'decoder.block{i + 1}', dch[i + 1])
    shapes['decoder.convout.weight'] = (2, dch[4], 3, 3)
    shapes['decoder.convout.bias'] = (2,)
    return shapes


def closed_form_state_dict(shapes, amplitude=0.25, dtype=torch.float32):
    """
    Deterministic seed-free weights: value = scale * sin(0.37*i + 1.3*key_index + 0.11),
    scale = amplitude / sqrt(fan) so that activations stay O(1) through the stack.
    """
    sd = {}
    for idx, (key, shape) in enumerate(shapes.items()):
        n = int(math.prod(shape))
        i = torch.arange(n, dtype=torch.float64)
        if key == 'skip_weights':
            v = 1.0 + 0.1 * torch.sin(0.9 * i + 0.3)
        else:
            fan = max(1, n // shape[0])
            v = amplitude / math.sqrt(fan) * torch.sin(0.37 * i + 1.3 * idx + 0.11)
            if key.endswith('weight'):
                v = v * 3.0
        sd[key] = v.reshape(shape).to(dtype)
    return sd


def residual_block(x, sd, prefix, dilation):
    """modules.py:755-777 : ELU(conv1x1(ELU(conv3x3_dil(x)))) + x, 'same' padding, dilation on H and T."""
    y = F.elu(F.conv2d(x, sd[f'{prefix}.conv1.0.weight'], sd[f'{prefix}.conv1.0.bias'],
                       padding=dilation, dilation=dilation))
    y = F.elu(F.conv2d(y, sd[f'{prefix}.conv2.0.weight'], sd[f'{prefix}.conv2.0.bias']))
    return y + x


def encoder_block(x, sd, prefix):
    """modules.py:632-655 : three residual blocks (d = 1, 2, 3) then Conv2d((4,1), stride (2,1)) + ELU."""
    y = residual_block(x, sd, f'{prefix}.block1', 1)
    y = residual_block(y, sd, f'{prefix}.block2', 2)
    y = residual_block(y, sd, f'{prefix}.block3', 3)
    return F.elu(F.conv2d(y, sd[f'{prefix}.sconv.0.weight'], sd[f'{prefix}.sconv.0.bias'], stride=(2, 1)))


def decoder_block(x, sd, prefix, output_padding):
    """modules.py:695-718 : ConvTranspose2d((4,1), stride (2,1), output_padding) + ELU then three residual blocks."""
    y = F.elu(F.conv_transpose2d(x, sd[f'{prefix}.tconv.0.weight'], sd[f'{prefix}.tconv.0.bias'],
                                 stride=(2, 1), output_padding=(output_padding, 0)))
    y = residual_block(y, sd, f'{prefix}.block1', 1)
    y = residual_block(y, sd, f'{prefix}.block2', 2)
    y = residual_block(y, sd, f'{prefix}.block3', 3)
    return y


def encoder_forward(coefficients, sd):
    """modules.py:448-483 : returns (latents (B,D,T), [5 embeddings])."""
    emb = [F.elu(F.conv2d(coefficients, sd['encoder.convin.0.weight'], sd['encoder.convin.0.bias'], padding=1))]
    for i in range(4):
        emb.append(encoder_block(emb[-1], sd, f'encoder.block{i + 1}'))
    latents = F.conv2d(emb[-1], sd['encoder.convlat.weight'], sd['encoder.convlat.bias']).squeeze(-2)
    return latents, emb


def decoder_forward(latents, sd, encoder_embeddings=None, feature_size=540):
    """modules.py:545-594 : latents (B,D+1,T) -> logits (B,2,F,T)."""
    _, padding = embedding_sizes(feature_size)
    y = F.elu(F.conv_transpose2d(latents.unsqueeze(-2), sd['decoder.convin.0.weight'], sd['decoder.convin.0.bias']))
    if encoder_embeddings is not None:
        y = y + encoder_embeddings[-1]
    for i in range(4):
        y = decoder_block(y, sd, f'decoder.block{i + 1}', padding[i])
        if encoder_embeddings is not None:
            y = y + encoder_embeddings[-2 - i]
    return F.conv2d(y, sd['decoder.convout.weight'], sd['decoder.convout.bias'], padding=1)


def apply_skip_connections(embeddings, sd):
    """modules.py:95-117."""
    if 'skip_weights' in sd and sd['skip_weights'] is not None:
        return [sd['skip_weights'][i] * e for i, e in enumerate(embeddings)]
    return None


def decode(latents, sd, embeddings=None, transcribe=False, feature_size=540):
    """modules.py:119-147 : append the indicator channel (1 = reconstruct, 0 = transcribe)."""
    indicator = (not transcribe) * torch.ones_like(latents[..., :1, :])
    return decoder_forward(torch.cat((latents, indicator), dim=-2), sd, embeddings, feature_size)


def forward(coefficients, sd, consistency=False, feature_size=540):
    """
    modules.py:338-393 with ``self.sliCQ(audio)`` already applied (the CQT is outside autograd,
    cqtwrapper.py:65).  Returns the reference 5 tensors (losses dict is always empty).
    """
    latents, emb = encoder_forward(coefficients, sd)
    emb = apply_skip_connections(emb, sd)
    reconstruction = decode(latents, sd, emb, False, feature_size)
    transcription = decode(latents, sd, emb, True, feature_size)
    if consistency:
        latents_trn, emb_trn = encoder_forward(transcription, sd)
        emb_trn = apply_skip_connections(emb_trn, sd)
        transcription_rec = decode(latents_trn, sd, emb_trn, False, feature_size)
        transcription_scr = decode(latents_trn, sd, emb_trn, True, feature_size)
    else:
        transcription_rec = transcription_scr = None
    return reconstruction, latents, transcription, transcription_rec, transcription_scr


def inference_coefficients(coefficients, sd, transcribe=False, feature_size=540):
    """modules.py:149-177 after the CQT."""
    with torch.no_grad():
        latents, emb = encoder_forward(coefficients, sd)
        emb = apply_skip_connections(emb, sd)
        return decode(latents, sd, emb, transcribe, feature_size)


def to_activations(coefficients):
    """modules.py:271-289 : tanh(||.||_2 over the channel dim)."""
    return torch.tanh(coefficients.norm(p=2, dim=-3))


def chunked_inference(audio, sd, cqt_forward, block_length, max_window_length, transcribe=False, feature_size=540):
    """
    modules.py:204-269.  ``cqt_forward(audio_chunk) -> (B,2,F,M)`` is the transform
    (torch tensor in/out); frame arithmetic restated exactly.
    """
    B = audio.size(0)
    audio = F.pad(audio, (0, -audio.size(-1) % block_length))
    hop = block_length // 2
    audio = F.pad(audio, [hop] * 2)
    n_chunks = (audio.size(-1) - hop) // hop
    M = max_window_length
    window = torch.signal.windows.hann(M, dtype=audio.dtype)
    n_frames = math.ceil((audio.size(-1) / block_length) * M)
    out = torch.zeros((B, 2, feature_size, n_frames), dtype=audio.dtype)
    for i in range(n_chunks):
        chunk = audio[..., i * hop: i * hop + block_length]
        o = inference_coefficients(cqt_forward(chunk), sd, transcribe, feature_size)
        fs = i * M // 2
        out[..., fs: fs + M] += window * o
    return out[..., M // 2: -M // 2]
