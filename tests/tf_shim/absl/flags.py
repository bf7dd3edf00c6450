class _Flags:
    def __getattr__(self, name):
        raise AttributeError(name)


FLAGS = _Flags()


def _define(name, default, help_=None, **kw):
    object.__setattr__(FLAGS, name, default)


DEFINE_string = DEFINE_integer = DEFINE_float = DEFINE_boolean = DEFINE_bool = _define
