"""Model registry lookup (reference: train.py:115-138,187-190; eval.py:67-70)."""
from . import frame_level_models, models, video_level_models


def find_class_by_name(name, modules=None):
    """Searches the provided modules for the named class and returns it (train.py:187-190)."""
    modules = modules or [frame_level_models, video_level_models]
    found = [getattr(module, name, None) for module in modules]
    try:
        return next(a for a in found if a)
    except StopIteration:
        raise ValueError("Unable to find model '%s'." % name)


def validate_class_name(flag_value, category="model", modules=None, expected_superclass=models.BaseModel):
    """train.py:115-138."""
    cls = find_class_by_name(flag_value, modules)
    if not issubclass(cls, expected_superclass):
        raise ValueError("%s '%s' doesn't inherit from %s." % (category, flag_value, expected_superclass.__name__))
    return True


def get_model(name):
    """``find_class_by_name(FLAGS.model, [...])()``: a no-arg-constructed model object (train.py:680-681)."""
    return find_class_by_name(name)()
