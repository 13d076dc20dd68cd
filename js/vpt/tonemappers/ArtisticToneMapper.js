'use strict';
// src/js/tonemappers/ArtisticToneMapper.js:10-85
const { AbstractToneMapper, P } = require('./AbstractToneMapper.js');
const { native } = require('../native.js');

class ArtisticToneMapper extends AbstractToneMapper {

static KIND() { return native().VPT_TONEMAPPER_ARTISTIC; }

constructor(gl, texture, options) {
    super(gl, texture, options);
    this.registerProperties([                                                               // :15-49
        { name: 'low', label: 'Low', type: 'spinner', value: 0 },
        { name: 'high', label: 'High', type: 'spinner', value: 1 },
        { name: 'mid', label: 'Midtones', type: 'slider', value: 0.5, min: 0.00001, max: 0.99999 },
        { name: 'saturation', label: 'Saturation', type: 'spinner', value: 1 },
        { name: 'gamma', label: 'Gamma', type: 'spinner', value: 2.2, min: 0 },
    ]);
}

_params() {                                                                                 // :75-79
    const p = super._params();
    p[P.LOW] = this.low; p[P.MID] = this.mid; p[P.HIGH] = this.high;
    p[P.SATURATION] = this.saturation; p[P.GAMMA] = this.gamma;
    return p;
}

}
module.exports = { ArtisticToneMapper };
