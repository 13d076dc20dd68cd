"""EventTarget / CustomEvent / PropertyBag — src/js/PropertyBag.js:1-16 plus the DOM event shims it relies on."""


class Event:
    def __init__(self, type_):
        self.type = type_


class CustomEvent(Event):
    def __init__(self, type_, init=None):
        super().__init__(type_)
        self.detail = (init or {}).get('detail')


class EventTarget:
    def __init__(self):
        self._listeners = {}

    def addEventListener(self, type_, listener):
        self._listeners.setdefault(type_, [])
        if listener not in self._listeners[type_]:
            self._listeners[type_].append(listener)

    def removeEventListener(self, type_, listener):
        if listener in self._listeners.get(type_, []):
            self._listeners[type_].remove(listener)

    def dispatchEvent(self, event):
        for listener in list(self._listeners.get(event.type, [])):
            listener(event)
        return True


class PropertyBag(EventTarget):
    """PropertyBag.js:1-16: ``properties`` registry; each registered property becomes an attribute."""

    def __init__(self):
        super().__init__()
        self.properties = []

    def registerProperties(self, properties):
        self.properties.extend(properties)
        for prop in properties:
            setattr(self, prop['name'], prop['value'])
