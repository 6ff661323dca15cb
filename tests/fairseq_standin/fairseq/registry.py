REGISTRIES = {}


def setup_registry(registry_name, base_class=None, default=None, required=False):
    key = registry_name[2:].replace("-", "_")
    table, class_names, dataclasses = {}, set(), {}
    REGISTRIES[key] = {"registry": table, "default": default, "dataclass_registry": dataclasses}

    def build_x(cfg, *args, **kwargs):
        choice = cfg if isinstance(cfg, str) else getattr(cfg, key, None)
        return table[choice](cfg, *args, **kwargs)

    def register_x(name, dataclass=None):
        def register_x_cls(cls):
            if name in table:
                raise ValueError("Cannot register duplicate {} ({})".format(key, name))
            if cls.__name__ in class_names:
                raise ValueError("Cannot register {} with duplicate class name ({})".format(key, cls.__name__))
            if base_class is not None and not issubclass(cls, base_class):
                raise ValueError("{} must extend {}".format(cls.__name__, base_class.__name__))
            table[name] = cls
            class_names.add(cls.__name__)
            return cls

        return register_x_cls

    return build_x, register_x, table, dataclasses
