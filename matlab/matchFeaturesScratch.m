function [matches, matchMetric] = matchFeaturesScratch(F1, F2, varargin)
    %MATCHFEATURESSCRATCH Shadows PP/featureMatching/matchFeaturesScratch.m for float descriptors.
    %   Name/value pairs as the reference (:59-76); 'Exhaustive' and the float 'Approximate' variants are all
    %   served by the exact device search.  Binary descriptors go through nearest2HammingExhaustiveMEX below.
    p = inputParser; p.KeepUnmatched = true;
    addParameter(p, 'Method', 'Exhaustive');
    addParameter(p, 'MatchThreshold', 3.5);
    addParameter(p, 'MaxRatio', 0.6);
    addParameter(p, 'Unique', true);
    parse(p, varargin{:});
    o = p.Results;
    opts = struct('MaxRatio', o.MaxRatio, 'MatchThreshold', o.MatchThreshold, 'Unique', double(o.Unique));
    [matches, matchMetric] = aps_mex('match_features', single(F1), single(F2), opts);
end
