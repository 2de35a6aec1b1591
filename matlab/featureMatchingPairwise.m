function matches = featureMatchingPairwise(input, allDescriptors, numImg)
    %FEATUREMATCHINGPAIRWISE Shadows PP/featureMatching/featureMatchingPairwise.m: all upper-triangular pairs
    %   in one batched device call (exhaustive 2-NN + ratio + threshold + unique), n x n cell of M x 2 double.
    arguments
        input struct
        allDescriptors cell
        numImg (1, 1) {mustBeNumeric, mustBeFinite, mustBePositive}
    end
    opts = struct('MaxRatio', input.Ratiothreshold, 'MatchThreshold', input.Matchingthreshold, 'Unique', 1);
    matches = aps_mex('match_pairwise', cellfun(@single, allDescriptors(1:numImg), 'UniformOutput', false), opts);
end
