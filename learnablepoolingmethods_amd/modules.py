"""Base class for modules (reference: modules.py:18-23)."""


class BaseModule(object):
    """Inherit from this class when implementing new modules."""

    def forward(self, unused_module_input, **unused_params):
        raise NotImplementedError()
