"""``@persistent_class`` in the reference (torch_utils/persistence.py:37-134) pickles the defining
module's source next to every instance.  That storage format is out of scope here;
this decorator keeps the *attributes* model code and training_loop.py read — ``init_args``,
``init_kwargs`` — and leaves pickling to the class's normal import path."""
import copy
import functools


def persistent_class(orig_class):
    orig_init = orig_class.__init__

    @functools.wraps(orig_init)
    def __init__(self, *args, **kwargs):
        if not hasattr(self, '_init_args'):            # outermost constructor call wins
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)
        orig_init(self, *args, **kwargs)

    orig_class.__init__ = __init__
    orig_class.init_args = property(lambda self: copy.deepcopy(self._init_args))
    orig_class.init_kwargs = property(lambda self: copy.deepcopy(self._init_kwargs))
    orig_class._p3d_persistent = True
    return orig_class


def is_persistent(obj):
    return bool(getattr(obj, '_p3d_persistent', False))


def import_hook(hook):          # API parity only; nothing to hook without source pickling
    return hook
