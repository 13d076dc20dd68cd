'use strict';
// src/js/tonemappers/ToneMapperFactory.js:12-27
const { ArtisticToneMapper } = require('./ArtisticToneMapper.js');
const { RangeToneMapper } = require('./RangeToneMapper.js');
const C = require('./CurveToneMappers.js');

function ToneMapperFactory(which) {
    switch (which) {
        case 'artistic': return ArtisticToneMapper;
        case 'range': return RangeToneMapper;
        case 'reinhard': return C.ReinhardToneMapper;
        case 'reinhard2': return C.Reinhard2ToneMapper;
        case 'uncharted2': return C.Uncharted2ToneMapper;
        case 'filmic': return C.FilmicToneMapper;
        case 'unreal': return C.UnrealToneMapper;
        case 'aces': return C.AcesToneMapper;
        case 'lottes': return C.LottesToneMapper;
        case 'uchimura': return C.UchimuraToneMapper;
        default: throw new Error('No suitable class');
    }
}
module.exports = { ToneMapperFactory };
