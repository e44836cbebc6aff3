"""Network factories of the synthesis path, under the reference's names
(/root/reference/stylegan_code_finder/networks/__init__.py):

* ``load_weights`` (:22-29)                      checkpoint dict -> ``network.load_state_dict`` (optional ``key``, ``strict``);
* ``get_stylegan2_generator`` (:36-41)           ``Generator(image_size, latent_size, n_mlp, channel_multiplier)`` + ``g_ema`` weights;
* ``get_swagan_generator`` (:354-362)            the same for the wavelet generator;
* ``get_autoencoder`` (:388-404) / ``load_autoencoder_or_generator`` (:415-423)
                                                 the object ``generate_images(batch, autoencoder, ...)`` and
                                                 ``build_latent_and_noise_generator(autoencoder, ...)`` are handed: only its
                                                 ``.decoder`` is on the hot path (utils/dataset_creation.py:36,50-57), so the
                                                 generator-only branch (no ``stylegan_checkpoint`` key in the config: weights
                                                 under ``'g_ema'``, strict) is what exists here.  The encoder families of the
                                                 projection research code (out of scope) are not rebuilt: asking
                                                 for the full-autoencoder branch raises ``NotImplementedError``.

Imports are deferred into the functions: ``networks`` is the package every model module lives in.
"""
import argparse
from pathlib import Path
from typing import Union

import torch
from torch import nn


def load_weights(network: nn.Module, model_file: Union[str, Path], *, key: str = None, strict: bool = True,
                 convert: bool = False) -> nn.Module:
    if convert:
        raise NotImplementedError("convert_autoencoder_checkpoint belongs to the encoder research code (out of scope)")
    weights = torch.load(model_file, map_location='cpu')
    if key is not None and key in weights:
        weights = weights[key]
    network.load_state_dict(weights, strict=strict)
    return network


def get_stylegan2_generator(image_size, latent_size, n_mlp=8, channel_multiplier=2, init_ckpt=None, ckpt_key='g_ema',
                            strict=True):
    from networks.stylegan2.model import Generator
    generator = Generator(image_size, latent_size, n_mlp, channel_multiplier=channel_multiplier)
    if init_ckpt is not None:
        load_weights(generator, init_ckpt, key=ckpt_key, strict=strict)
    return generator


def get_swagan_generator(image_size, latent_size, n_mlp=8, channel_multiplier=2, init_ckpt=None, ckpt_key='g_ema',
                         strict=True):
    from networks.swagan.model import Generator
    generator = Generator(image_size, latent_size, n_mlp, channel_multiplier=channel_multiplier)
    if init_ckpt is not None:
        load_weights(generator, init_ckpt, key=ckpt_key, strict=strict)
    return generator


class StyleganAutoencoder(nn.Module):
    """The (encoder, decoder) holder of networks/encoder/autoencoder.py:13-19 with the decoder only."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.use_generated_noise = True

    def encode(self, x):
        if self.encoder is None:
            raise NotImplementedError("this autoencoder holds a generator only (image encoding is out of scope)")
        return self.encoder(x)

    def forward(self, x):
        latent_codes = self.encode(x)
        image, _ = self.decoder([latent_codes.latent], input_is_latent=latent_codes.latent.dim() == 3,
                                noise=latent_codes.noise)
        return image


def get_autoencoder(config: dict, init_ckpt: str = None) -> StyleganAutoencoder:
    assert config['stylegan_variant'] in [1, 2, 'swagan'], "Stylegan Variant Unknown"
    if config['stylegan_variant'] == 1:
        raise NotImplementedError("StyleGAN1 is not on the MI355X hot path (SURVEY.md §2)")
    make = get_swagan_generator if config['stylegan_variant'] == 'swagan' else get_stylegan2_generator
    generator = make(config['image_size'], config['latent_size'], n_mlp=config.get('n_mlp', 8),
                     channel_multiplier=config.get('channel_multiplier', 2), init_ckpt=init_ckpt, strict=False)
    return StyleganAutoencoder(None, generator)


def load_autoencoder_or_generator(args: argparse.Namespace, config: dict) -> StyleganAutoencoder:
    autoencoder = get_autoencoder(config).to(args.device)
    # the reference decides by this key whether the checkpoint holds a full autoencoder or just the generator
    if 'stylegan_checkpoint' in config:
        raise NotImplementedError("full autoencoder checkpoints need the encoder networks (out of scope); "
                                  "generator checkpoints ('g_ema') are supported")
    autoencoder.decoder = load_weights(autoencoder.decoder, args.checkpoint, key='g_ema')
    return autoencoder
