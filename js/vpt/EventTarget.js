'use strict';
// Minimal EventTarget / Event / CustomEvent for Node 12 (the reference relies on the browser's: PropertyBag.js:1,
// Component.js:1, Application.js:130-137).
class Event { constructor(type) { this.type = type; } }
class CustomEvent extends Event {
    constructor(type, init) { super(type); this.detail = init ? init.detail : undefined; }
}
class EventTarget {
    constructor() { this._listeners = Object.create(null); }
    addEventListener(type, listener) {
        const list = this._listeners[type] || (this._listeners[type] = []);
        if (list.indexOf(listener) < 0) { list.push(listener); }
    }
    removeEventListener(type, listener) {
        const list = this._listeners[type];
        if (!list) { return; }
        const i = list.indexOf(listener);
        if (i >= 0) { list.splice(i, 1); }
    }
    dispatchEvent(event) {
        const list = this._listeners[event.type];
        if (list) { for (const l of list.slice()) { l.call(this, event); } }
        return true;
    }
}
module.exports = { Event, CustomEvent, EventTarget };
