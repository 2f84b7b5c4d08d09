"""Parser registry entries (the Go packages register themselves from init(); importing this
package does the same: scm, scm+, idm, netidm, r900)."""
from . import scm, idm, r900  # noqa: F401
