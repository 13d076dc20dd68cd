'use strict';
// src/js/PropertyBag.js:1-16
const { EventTarget } = require('./EventTarget.js');

class PropertyBag extends EventTarget {

constructor() {
    super();
    this.properties = [];
}

registerProperties(properties) {
    this.properties.push(...properties);
    for (const property of properties) {
        this[property.name] = property.value;
    }
}

}
module.exports = { PropertyBag };
