"""Model hyper-parameters of the reference configs as plain dicts (the ml_collections / absl flag
machinery is out of scope).  Each entry is the ``model_config`` of the cited reference file."""

RD_LAMBDAS = [0.08, 0.02, 0.005, 0.00125, 0.04, 0.01, 0.0025]     # get_hyper() sweeps (two_layer_syn.py:73)


def two_layer_syn(rd_lambda=0.08):
    """mshyper/configs/two_layer_syn.py:28-46 -- ELIC analysis + 2-layer residual synthesis."""
    return dict(
        rd_lambda=rd_lambda,
        transform_config=dict(
            analysis=dict(cls="ElicAnalysis", channels=(192, 192, 192, 320)),
            synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                           activation_type="igdn", res_type="conv")),
        latent_config=dict(uq=dict(method="unoise")))


def jpegl(rd_lambda=0.08):
    """mshyper/configs/jpegl.py:28-45 -- ELIC analysis + JPEG-like one-layer synthesis (k=18, s=16)."""
    return dict(
        rd_lambda=rd_lambda,
        transform_config=dict(
            analysis=dict(cls="ElicAnalysis", channels=(192, 192, 192, 320)),
            synthesis=dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16)),
        latent_config=dict(uq=dict(method="unoise")))


def mbt2018(rd_lambda=0.08):
    """mshyper/configs/mbt2018.py:27-41 -- Minnen 2018 mean-scale hyperprior, 192 / 320 channels."""
    return dict(
        rd_lambda=rd_lambda,
        transform_config=dict(
            analysis=dict(cls="MBT2018Analysis", channels_base=192, output_channels=320),
            synthesis=dict(cls="MBT2018Synthesis", channels_base=192, output_channels=3)))


def two_layer_syn2(rd_lambda=0.08, hidden_channels=12):
    """mshyper/configs/two_layer_syn2.py:39-59 -- CNN analysis (256 -> 320) + 2-layer synthesis, mixedq."""
    return dict(
        rd_lambda=rd_lambda,
        transform_config=dict(
            analysis=dict(cls="CNNAnalysis", channels_base=256, output_channels=320),
            synthesis=dict(cls="TwoLayerSynthesis", channels=(hidden_channels, 3), strides=(8, 2), kernel_sizes=(13, 5),
                           activation_type="igdn")),
        latent_config=dict(uq=dict(method="mixedq")),
        offset_heuristic=False)


def itinf():
    """mshyper/configs/itinf.py:30-45 -- SGA overrides applied on top of a trained model's config."""
    return dict(
        scheduled_num_steps=3000,
        optimizer_config=dict(learning_rate=5e-3, reduce_lr_after=0.9, reduce_lr_factor=0.1, global_clipnorm=None,
                              warmup_until=0.0),
        latent_config=dict(uq=dict(method="sga", tau_r=5e-4, tau_ub=0.5, tau_t0=200)),
        offset_heuristic=False)


def bls2017(rd_lambda=0.08):
    """factorized/configs/bls2017.py:30-38 -- Balle 2017 factorized prior, 256 filters."""
    return dict(
        rd_lambda=rd_lambda,
        transform_config=dict(
            analysis=dict(cls="BLS2017Analysis", num_filters=256),
            synthesis=dict(cls="BLS2017Synthesis", num_filters=256)))


CONFIGS = dict(two_layer_syn=two_layer_syn, jpegl=jpegl, mbt2018=mbt2018, two_layer_syn2=two_layer_syn2,
               bls2017=bls2017)
