"""Registry helper (reference common/utils.py:58-71)."""


class ClassBuilder(dict):
    """``ClassBuilder({'A': A}).build('A', x=1) -> A(x=1)`` -- the transform plugin point."""

    def build(self, class_name, **kwargs):
        cls = self[class_name]
        return cls(**kwargs)
