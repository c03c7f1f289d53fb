"""sgpo_amd: MI355X-native implementation of SGPO's guided reverse-diffusion sampling loop."""


def install_dropin():
    """Make `models.*`, `sampling.*`, `problem.*` import the HIP-backed mirrors."""
    from .dropin import install
    return install()
