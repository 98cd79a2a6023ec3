"""gitcap: MI355X-native GIT-style video-caption inference behind the call surface of
farazali7/real-time-video-captioning (see DESIGN.md / INTEGRATION.md)."""
from .config import GitCapConfig, git_base, git_large, git_tiny          # noqa: F401
from .weights import synthetic_weights, canonical_shapes                  # noqa: F401


def __getattr__(name):
    # GitCaptioner pulls in torch + the HIP library; keep `import gitcap` light for host-only tools
    if name == "GitCaptioner":
        from .model import GitCaptioner
        return GitCaptioner
    raise AttributeError(name)
