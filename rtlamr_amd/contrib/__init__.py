"""Optional host-side companions of the hot path -- NOT part of it (the parsers are downstream of
Decoder.Decode and out of scope; with the cgo binding the reference's own Go parsers run unchanged).

`parsers/`: Python mirrors of the reference's scm / scmplus / idm / netidm / r900 / r900bcd packages and their CSV /
plain-text record forms.  They exist so that planted packets can be followed end to end in tests, in smoke() and before
bench.py's timed region, and so that `Decoder.EnableValidation` can read each parser's checksum rule (VALIDATOR).  The
C ABI (include/amrdemod.h, libamrdemod.so) does not know them.  tests/test_ref_translated.py holds them to the messages
of the reference's own parser sources (oracle/_ref)."""
