"""rtlamr_amd -- MI355X (gfx950) implementation of rtlamr's protocol.Decoder hot path.

The compute path is the HIP library rtlamr_amd/csrc/libamrdemod.so behind the C ABI of
include/amrdemod.h.  This package is the thin host-side mirror of the reference's Go API
(package protocol: Decoder / Parser / Data / PacketConfig) on top of that ABI.  There is no
CPU fallback: importing works anywhere, creating a Decoder needs a gfx950 GPU and the built
library, and fails loudly otherwise.
"""
from .protocol import (Data, Decoder, Message, PacketConfig, Parser, new_data, new_decoder,  # noqa: F401
                       new_parser, next_power_of_2, register_parser)
from .contrib import parsers  # noqa: F401  (registers scm, scm+, idm, netidm, r900, r900bcd configs)

__version__ = "0.1.0"
